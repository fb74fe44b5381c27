"""Convolution modules of the DEAL-YOLO graph: Conv, LDConv, Concat (drop-in for reference nn/modules/conv.py).

Parameters live in ordinary ``nn.Conv2d`` / ``nn.BatchNorm2d`` holders so that ``state_dict`` keys, default
initialisation and pickled reference checkpoints line up; the arithmetic is issued through
libdealyolo_hip.so -- the holders' own ``forward`` is never called.
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn

from ...hip import DY_ACT_NONE, DY_ACT_SILU
from ...hip.runtime import HipModule

__all__ = ("Conv", "LDConv", "Concat", "autopad")
# DY_SPLIT_COUT=0: one launch per fused conv whatever its output width (the last 64-wide cout group zero-padded)
SPLIT_COUT = __import__("os").environ.get("DY_SPLIT_COUT", "1") != "0"


def autopad(k, p=None, d=1):
    """'same' padding (reference nn/modules/conv.py:32-38)."""
    if d > 1:
        k = d * (k - 1) + 1 if isinstance(k, int) else [d * (x - 1) + 1 for x in k]
    if p is None:
        p = k // 2 if isinstance(k, int) else [x // 2 for x in k]
    return p


class Conv(HipModule):
    """conv(bias-free) -> BatchNorm2d -> SiLU; args (c1, c2, k, s, p, g, d, act) as reference nn/modules/conv.py:41-59."""
    default_act = nn.SiLU()

    def __init__(self, c1, c2, k=1, s=1, p=None, g=1, d=1, act=True):
        super().__init__()
        if g != 1 or d != 1 or k not in (1, 3) or s not in (1, 2) or (k == 1 and s != 1):
            raise NotImplementedError(f"Conv(k={k}, s={s}, g={g}, d={d}) is outside the HIP hot path (k in 1|3, s in 1|2, g=d=1)")
        self.conv = nn.Conv2d(c1, c2, k, s, autopad(k, p, d), groups=g, dilation=d, bias=False)
        self.bn = nn.BatchNorm2d(c2)
        self.act = self.default_act if act is True else act if isinstance(act, nn.Module) else nn.Identity()
        if not isinstance(self.act, (nn.SiLU, nn.Identity)):
            raise NotImplementedError("only SiLU / Identity activations are implemented in the HIP conv epilogue")

    def _act_code(self):
        return DY_ACT_SILU if isinstance(self.act, nn.SiLU) else DY_ACT_NONE

    def _build_specs(self, rt):
        rt.make_spec(id(self), self.conv, getattr(self, "bn", None), self._act_code(), self.conv.kernel_size[0],
                     self.conv.stride[0], name="Conv")
        self.__dict__["_split"] = None
        c2, k = self.conv.out_channels, self.conv.kernel_size[0]
        if SPLIT_COUT and not hasattr(self, "bn") and self.conv.bias is not None and k == 3 and self.conv.stride[0] == 1 and c2 > 64 and c2 % 64 in (16, 32):
            # fused (eval) 3x3 conv whose output width is 64 m + 16 / 32 (Detect's class branch for nc = 80: 80 = 64 + 16 channels): the conv
            # kernel's cout groups are 64 wide, so the last group would multiply 48 / 32 rows of zero weights -- as much matrix work for
            # 16 outputs as for 64.  The tail runs as a launch of its own on a 16- / 32-row group: two parameter views, two packs.
            from types import SimpleNamespace as NS
            lo = c2 // 64 * 64
            w, b = self.conv.weight.data, self.conv.bias.data
            parts = [NS(weight=NS(data=w[:lo]), bias=NS(data=b[:lo])), NS(weight=NS(data=w[lo:]), bias=NS(data=b[lo:]))]
            self.__dict__["_split"] = (lo, [rt.make_spec((id(self), "cout", i), pc, None, self._act_code(), k, 1, name=f"Conv.cout{i}")
                                            for i, pc in enumerate(parts)])

    def out_hw(self, h, w):
        k, s = self.conv.kernel_size[0], self.conv.stride[0]
        p = k // 2
        return (h + 2 * p - k) // s + 1, (w + 2 * p - k) // s + 1

    def forward_act(self, x, out=None, res=None, defer_apply=False):
        eng = self.rt.eng
        spec = self.rt.spec(self)
        if hasattr(self, "bn"):
            return eng.conv_bn_act(spec, x, out, res, defer_apply=defer_apply)
        split = self.__dict__.get("_split")
        if split is not None and res is None and eng.tape is None:
            lo, (sa, sb) = split
            from ...hip.engine import Act
            x = eng.dense(x)
            Ho, Wo = eng.out_hw(spec, x)
            y = out if out is not None else eng.new_act(x.N, Ho, Wo, spec.cout)
            eng.conv_fused(sa, x, Act(y.st, y.c0, lo, y.needs_grad))
            eng.conv_fused(sb, x, Act(y.st, y.c0 + lo, spec.cout - lo, y.needs_grad))
            return y
        return eng.conv_fused(spec, x, out, res)

    forward_fuse = HipModule.forward


class Concat(HipModule):
    """Channel concatenation (reference nn/modules/conv.py:338-348)."""

    def __init__(self, dimension=1):
        super().__init__()
        self.d = dimension
        if dimension != 1:
            raise NotImplementedError("only channel concatenation is on the hot path")

    def forward_act(self, xs, out=None):
        return self.rt.eng.concat(xs)


class LDConv(HipModule):
    """Linear deformable convolution (reference nn/modules/conv.py:350-503): offsets from a 3x3 conv, N bilinear samples
    per output pixel, (N,1) column conv -> BN -> SiLU."""

    def __init__(self, inc, outc, num_param, stride=1, bias=None):
        super().__init__()
        self.num_param, self.stride = num_param, stride
        self.conv = nn.Sequential(nn.Conv2d(inc, outc, kernel_size=(num_param, 1), stride=(num_param, 1), bias=bias),
                                  nn.BatchNorm2d(outc), nn.SiLU())
        self.p_conv = nn.Conv2d(inc, 2 * num_param, kernel_size=3, padding=1, stride=stride)
        nn.init.constant_(self.p_conv.weight, 0)
        self.register_buffer("p_n", self._get_p_n(num_param))

    @staticmethod
    def _get_p_n(N):
        base = round(math.sqrt(N))
        rows, mod = N // base, N % base
        px = [r for r in range(rows) for _ in range(base)] + [rows] * mod
        py = [c for _ in range(rows) for c in range(base)] + list(range(mod))
        return torch.tensor(px + py, dtype=torch.int64).view(1, 2 * N, 1, 1)

    def out_hw(self, h, w):
        s = self.stride
        return (h + 2 - 3) // s + 1, (w + 2 - 3) // s + 1

    def _build_specs(self, rt):
        inc = self.p_conv.in_channels
        cphys = (inc + 7) // 8 * 8
        rt.make_spec((id(self), "p_conv"), self.p_conv, None, DY_ACT_NONE, 3, self.stride, name="LDConv.p_conv")
        rt.make_spec((id(self), "conv", cphys), self.conv[0], self.conv[1], DY_ACT_SILU, 1, 1, name="LDConv.conv",
                     ld=(self.num_param, cphys, inc))
        self.__dict__["_pn_i32"] = self.p_n.reshape(-1).to(torch.int32).to(rt.eng.device).contiguous()

    def forward_act(self, x, out=None):
        return self.rt.ldconv(self, x, out)
