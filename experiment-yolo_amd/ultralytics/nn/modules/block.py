"""Block modules of the DEAL-YOLO graph: DFL, SPPF, C2f, Bottleneck (drop-in for reference nn/modules/block.py)."""
from __future__ import annotations

import torch
import torch.nn as nn

from ...hip.runtime import HipModule
from .conv import Conv

__all__ = ("DFL", "SPPF", "C2f", "Bottleneck")


class DFL(nn.Module):
    """Integral of the distribution-focal-loss bins (reference nn/modules/block.py:37-55).  Only a parameter holder here:
    the frozen arange 'conv' is applied inside the decode / loss kernels."""

    def __init__(self, c1=16):
        super().__init__()
        self.conv = nn.Conv2d(c1, 1, 1, bias=False).requires_grad_(False)
        self.conv.weight.data[:] = torch.arange(c1, dtype=torch.float).view(1, c1, 1, 1)
        self.c1 = c1


class Bottleneck(HipModule):
    """3x3 -> 3x3 with optional shortcut (reference nn/modules/block.py:320-335)."""

    def __init__(self, c1, c2, shortcut=True, g=1, k=(3, 3), e=0.5):
        super().__init__()
        c_ = int(c2 * e)
        k0 = k[0][0] if isinstance(k[0], (tuple, list)) else k[0]
        k1 = k[1][0] if isinstance(k[1], (tuple, list)) else k[1]
        self.cv1 = Conv(c1, c_, k0, 1)
        self.cv2 = Conv(c_, c2, k1, 1, g=g)
        self.add = shortcut and c1 == c2

    def forward_act(self, x, out=None):
        return self.cv2.forward_act(self.cv1.forward_act(x), out, x if self.add else None)


class C2f(HipModule):
    """CSP bottleneck with two convolutions (reference nn/modules/block.py:209-232).  chunk/cat are free: cv1 writes its
    two halves and every Bottleneck its output straight into channel slices of ONE buffer that cv2 then reads."""

    def __init__(self, c1, c2, n=1, shortcut=False, g=1, e=0.5):
        super().__init__()
        self.c = int(c2 * e)
        self.cv1 = Conv(c1, 2 * self.c, 1, 1)
        self.cv2 = Conv((2 + n) * self.c, c2, 1)
        self.m = nn.ModuleList(Bottleneck(self.c, self.c, shortcut, g, k=((3, 3), (3, 3)), e=1.0) for _ in range(n))

    def forward_act(self, x, out=None):
        eng, c, n = self.rt.eng, self.c, len(self.m)
        from ...hip.engine import PLANAR, SegAct
        if PLANAR and out is None:
            # the concatenation is never built: cv1's output (both halves) and every Bottleneck's output are tensors of their own -- the
            # kernels that touch one member alone (BatchNorm apply / backward reduce, the 3x3 convs) walk contiguous memory -- and cv2,
            # a 1x1 conv, reads / back-propagates through a segment table (reference block.py:222-226: y = list(cv1(x).chunk(2, 1)) ...);
            # a shape the segmented kernels do not take is copied together by conv_bn_act (Engine.dense)
            spec = self.rt.spec(self.cv1)
            if eng.planes_ok(spec, x):  # both halves of the chunk as tensors of their own
                y0, prev = eng.new_planes(x.N, x.H, x.W, c)
                self.cv1.forward_act(x, SegAct([y0, prev]))
                parts = [y0, prev]
            else:
                y01 = self.cv1.forward_act(x)
                parts, prev = [y01], y01.sub(c, c)
            for b in self.m:
                prev = b.forward_act(prev)
                parts.append(prev)
            return self.cv2.forward_act(SegAct(parts))
        cat = eng.new_storage(x.N, x.H, x.W, (2 + n) * c)
        self.cv1.forward_act(x, cat.act(0, 2 * c))
        prev = cat.act(c, c)
        for j, b in enumerate(self.m):
            prev = b.forward_act(prev, cat.act((2 + j) * c, c))
        return self.cv2.forward_act(cat.act(), out)

    forward_split = HipModule.forward


class SPPF(HipModule):
    """Spatial pyramid pooling - fast (reference nn/modules/block.py:151-171)."""

    def __init__(self, c1, c2, k=5):
        super().__init__()
        if k != 5:
            raise NotImplementedError("SPPF: only k=5 pooling is implemented on the HIP path")
        c_ = c1 // 2
        self.cv1 = Conv(c1, c_, 1, 1)
        self.cv2 = Conv(c_ * 4, c2, 1, 1)
        self.m = nn.MaxPool2d(kernel_size=k, stride=1, padding=k // 2)

    def forward_act(self, x, out=None):
        eng = self.rt.eng
        c_ = self.cv1.conv.out_channels
        cat = eng.new_storage(x.N, x.H, x.W, 4 * c_)
        self.cv1.forward_act(x, cat.act(0, c_))
        eng.sppf_pools(cat, c_)
        return self.cv2.forward_act(cat.act(), out)
