"""Detect head (drop-in for reference nn/modules/head.py:19-87)."""
from __future__ import annotations

import math
import os

import torch
import torch.nn as nn

from ...hip import DY_ACT_NONE
from ...hip.runtime import HipModule
from .block import DFL
from .conv import Conv

__all__ = ("Detect",)


class HeadOut:
    """Raw head outputs of one forward: per level fp32 (B,H,W,64) DFL logits and (B,H,W,ncp) class logits (ncp = nc
    rounded up to 8), plus lazily created fp16 gradient buffers of the same shapes (filled by the loss kernels)."""

    def __init__(self, box, cls, nc, strides):
        self.box, self.cls, self.nc, self.strides = box, cls, nc, strides
        self.dbox = self.dcls = None
        self.fill_box = None  # inside a StepPlan trace the box logits are not written (dy_head_box_decode); callable that writes them
        self.infer = None     # eval forward with the fused inference tail (Detect._infer_tail): callable -> y (B, 4+nc, A); box / cls
                              # hold None until somebody asks for the logits (materialize)

    def materialize(self):
        """Write the box logits if the forward left them out (the recorded training step never reads them): an eager launch of the
        plain final convs over the activations still resident from the last step."""
        if self.fill_box is not None:
            self.fill_box()

    def alloc_grads(self):
        if self.dbox is None:
            self.dbox = [torch.zeros(b.shape, dtype=torch.float16, device=b.device) for b in self.box]
            self.dcls = [torch.zeros(c.shape, dtype=torch.float16, device=c.device) for c in self.cls]

    def as_reference_list(self):
        """[(B, no, H, W)] fp32 views/copies in the reference's training-output format (head.py:45-48)."""
        self.materialize()
        return [torch.cat((b, c[..., : self.nc]), -1).permute(0, 3, 1, 2) for b, c in zip(self.box, self.cls)]


class LazyFeats:
    """List-like view of the reference-format per-level feature maps; the cat/permute copies happen on first access."""

    def __init__(self, ho):
        self._ho, self._list = ho, None

    def _get(self):
        if self._list is None:
            self._list = self._ho.as_reference_list()
        return self._list

    def __len__(self):
        return len(self._ho.box)

    def __getitem__(self, i):
        return self._get()[i]

    def __iter__(self):
        return iter(self._get())


class Detect(HipModule):
    """YOLOv8 Detect head: per level cv2 = Conv3x3 -> Conv3x3 -> Conv2d1x1(64), cv3 = ... -> Conv2d1x1(nc)."""
    dynamic = False
    export = False
    shape = None
    anchors = torch.empty(0)
    strides = torch.empty(0)

    def __init__(self, nc=80, ch=()):
        super().__init__()
        self.nc = nc
        self.nl = len(ch)
        self.reg_max = 16
        self.no = nc + self.reg_max * 4
        self.stride = torch.zeros(self.nl)
        c2, c3 = max((16, ch[0] // 4, self.reg_max * 4)), max(ch[0], min(self.nc, 100))
        self.cv2 = nn.ModuleList(nn.Sequential(Conv(x, c2, 3), Conv(c2, c2, 3), nn.Conv2d(c2, 4 * self.reg_max, 1)) for x in ch)
        self.cv3 = nn.ModuleList(nn.Sequential(Conv(x, c3, 3), Conv(c3, c3, 3), nn.Conv2d(c3, self.nc, 1)) for x in ch)
        self.dfl = DFL(self.reg_max) if self.reg_max > 1 else nn.Identity()

    def _build_specs(self, rt):
        for br in ("cv2", "cv3"):
            for l, seq in enumerate(getattr(self, br)):
                rt.make_spec((id(self), br, l), seq[2], None, DY_ACT_NONE, 1, 1, name=f"Detect.{br}.{l}.2")

    def bias_init(self):
        """Reference head.py:76-83 (requires stride)."""
        for a, b, s in zip(self.cv2, self.cv3, self.stride):
            a[-1].bias.data[:] = 1.0
            b[-1].bias.data[: self.nc] = math.log(5 / self.nc / (640 / s) ** 2)

    def _infer_tail(self, xs):
        """Eval forward asked for by an InferPlan (``eng.infer_head``): the conv stacks of every level, then NOTHING -- the two final
        convs, the DFL decode and the sigmoid are one dy_head_infer_levels launch (``ho.infer()``) that the caller issues with a fresh
        output tensor per call; the fp32 logits are only written when somebody asks for them (``ho.materialize()``)."""
        import ctypes as C
        rt, eng = self.rt, self.rt.eng
        n, nb, ncp = len(xs), 4 * self.reg_max, (self.nc + 7) // 8 * 8
        bsp = [rt.specs[(id(self), "cv2", l)] for l in range(n)]
        csp = [rt.specs[(id(self), "cv3", l)] for l in range(n)]
        if not (len({(sp.cin, sp.cout) for sp in csp}) == 1 and all(sp.ks == 1 and sp.ld is None and sp.bias is not None for sp in bsp + csp)
                and all(eng.L.dy_head_infer_supported(sp.cin, sp.cout, csp[0].cin, self.nc) for sp in bsp)):
            return None
        a = [self.cv2[l][1].forward_act(self.cv2[l][0].forward_act(x)) for l, x in enumerate(xs)]
        c = [self.cv3[l][1].forward_act(self.cv3[l][0].forward_act(x)) for l, x in enumerate(xs)]
        ho = HeadOut([None] * n, [None] * n, self.nc, [float(s) for s in self.stride])
        B, A, dev = xs[0].N, sum(x.H * x.W for x in xs), eng.device
        P, I, F = C.c_void_p, C.c_int, C.c_float
        args = (n, eng._arr(P, [t.ptr for t in a]), eng._arr(I, [t.ld for t in a]), eng._arr(P, [sp.weight.data_ptr() for sp in bsp]),
                eng._arr(P, [sp.bias.data_ptr() for sp in bsp]), eng._arr(P, [t.ptr for t in c]), eng._arr(I, [t.ld for t in c]),
                eng._arr(P, [sp.weight.data_ptr() for sp in csp]), eng._arr(P, [sp.bias.data_ptr() for sp in csp]),
                eng._arr(I, [x.H for x in xs]), eng._arr(I, [x.W for x in xs]), eng._arr(F, ho.strides), B, csp[0].cin, self.nc)

        def infer(y=None):
            from ...hip import check
            if y is None:
                y = torch.empty((B, 4 + self.nc, A), dtype=torch.float32, device=dev)
            check(eng.L.dy_head_infer_levels(*args, y.data_ptr(), eng.stream), "dy_head_infer_levels")
            return y

        def fill():  # the logits themselves, for callers that index the per-level feature maps (reference head.py:74 returns them too)
            from ...hip import DY_EPI_BIAS, DY_EPI_F32OUT
            assert eng.rec is None, "logits can only be materialised outside a trace"
            for l, x in enumerate(xs):
                if ho.box[l] is None:
                    ho.box[l] = torch.empty((x.N, x.H, x.W, nb), dtype=torch.float32, device=dev)
                    ho.cls[l] = torch.zeros((x.N, x.H, x.W, ncp), dtype=torch.float32, device=dev)
                eng._conv_raw(bsp[l], a[l], ho.box[l].data_ptr(), nb, DY_EPI_BIAS | DY_EPI_F32OUT, 0, bsp[l].bias)
                eng._conv_raw(csp[l], c[l], ho.cls[l].data_ptr(), ncp, DY_EPI_BIAS | DY_EPI_F32OUT, 0, csp[l].bias)
        ho.infer, ho.fill_box = infer, fill
        return ho

    def forward_act(self, xs, out=None):
        rt = self.rt
        eng = rt.eng
        if eng.infer_head and eng.tape is None and not eng.training:
            ho = self._infer_tail(xs)
            if ho is not None:
                return ho
        ncp = (self.nc + 7) // 8 * 8
        nb = 4 * self.reg_max
        boxes = [eng.transient((x.N, x.H, x.W, nb), torch.float32) for x in xs]
        # the 1x1 class conv writes channels [0, nc): padding channels (nc rounded up to 8) need a defined value only if they exist
        clss = [eng.transient((x.N, x.H, x.W, ncp), torch.float32) if ncp == self.nc else
                torch.zeros((x.N, x.H, x.W, ncp), dtype=torch.float32, device=eng.device) for x in xs]
        eng.hold(*boxes, *clss)
        ho = HeadOut(boxes, clss, self.nc, [float(s) for s in self.stride])
        if eng.tape is not None:
            ho.alloc_grads()
            eng.hold(*ho.dbox, *ho.dcls)
        from ...hip.engine import HEAD_APPLY, HEAD_DECODE
        # inside a StepPlan trace, when every level's box conv qualifies: forward fused with the loss's decode, backward from rows
        fused = HEAD_DECODE and eng.pending_decode is not None and all(eng.rows_capable(rt.specs[(id(self), "cv2", l)]) for l in range(len(xs)))
        lazy = []
        from ...hip.engine import HEAD_BATCH
        cspecs = [rt.specs[(id(self), "cv3", l)] for l in range(len(xs))]
        cls_all = HEAD_BATCH and all(eng.cls_capable(sp, ncp) for sp in cspecs) and len({(sp.cin, sp.cout) for sp in cspecs}) == 1
        box_items, box_fns, cls_items = [], [], []
        from ...hip.engine import BN_GROUP, HEAD_STREAMS
        staged = None
        if HEAD_STREAMS and eng.pending_decode is not None and eng.tape is not None and eng.cur_sid == 0:
            # inside a StepPlan trace: every level's two conv stacks are an independent chain until the loss -- level l > 0 is recorded
            # on side stream l, so the small levels' launches (50-200 workgroups) run beside the 160x160 level's and beside each other
            staged = [None] * (2 * len(xs))
            order = os.environ.get("DY_HEAD_STREAMS_ORDER", "")
            first_alone = order == "after"  # level 0 (chip-filling) on its own first, then the small levels beside each other
            if first_alone:
                staged[0] = self.cv2[0][1].forward_act(self.cv2[0][0].forward_act(xs[0]), defer_apply=fused and HEAD_APPLY)
                staged[1] = self.cv3[0][1].forward_act(self.cv3[0][0].forward_act(xs[0]), defer_apply=eng.cls_capable(cspecs[0], ncp) and HEAD_APPLY)
            for l in range(1, len(xs)):
                eng.fork_branch(l)
            for l in (range(1 if first_alone else 0, len(xs)) if order != "rev" else reversed(range(len(xs)))):
                with eng.branch(l):
                    staged[2 * l] = self.cv2[l][1].forward_act(self.cv2[l][0].forward_act(xs[l]), defer_apply=fused and HEAD_APPLY)
                    staged[2 * l + 1] = self.cv3[l][1].forward_act(self.cv3[l][0].forward_act(xs[l]),
                                                                   defer_apply=eng.cls_capable(cspecs[l], ncp) and HEAD_APPLY)
            for l in range(1, len(xs)):
                eng.join_branch(l)
        elif BN_GROUP and eng.pending_decode is not None and eng.tape is not None and 2 * len(xs) <= eng.L.dy_bn_group_max():
            # inside a StepPlan trace the two stages of the six branches run as groups: six conv launches, ONE apply launch (and ONE
            # backward reduce launch) per stage instead of six -- the 80x80 / 40x40 levels' passes run beside the 160x160 level's
            first = eng.conv_bn_act_group([(rt.spec(m[l][0]), x) for l, x in enumerate(xs) for m in (self.cv2, self.cv3)])
            defer = [(fused if b == 0 else eng.cls_capable(cspecs[l], ncp)) and HEAD_APPLY for l in range(len(xs)) for b in (0, 1)]
            staged = eng.conv_bn_act_group([(rt.spec(m[l][1]), first[2 * l + b]) for l in range(len(xs)) for b, m in enumerate((self.cv2, self.cv3))],
                                           defer=defer)
        for l, x in enumerate(xs):
            # fused: ``a`` is read by dy_head_box_decode, the rows backward and the loss only -- all of which apply BatchNorm + SiLU
            # themselves, so the Conv in front leaves its apply launch out
            cspec = cspecs[l]
            cls_fused = eng.cls_capable(cspec, ncp)
            if staged is not None:
                a, c = staged[2 * l], staged[2 * l + 1]
            else:
                a = self.cv2[l][1].forward_act(self.cv2[l][0].forward_act(x), defer_apply=fused and HEAD_APPLY)
                c = self.cv3[l][1].forward_act(self.cv3[l][0].forward_act(x), defer_apply=cls_fused and HEAD_APPLY)
            if fused:
                if HEAD_BATCH:  # all levels in one launch per kind, issued after the loop
                    box_items.append((rt.specs[(id(self), "cv2", l)], a, l))
                    box_fns.append(lambda l=l: (ho.dbox[l].data_ptr(), nb))
                else:
                    eng.conv_bias_decode(rt.specs[(id(self), "cv2", l)], a, lambda l=l: (ho.dbox[l].data_ptr(), nb), l)
                lazy.append((rt.specs[(id(self), "cv2", l)], a, boxes[l], eng.unapplied(a)))
            else:
                eng.conv_bias(rt.specs[(id(self), "cv2", l)], a, boxes[l].data_ptr(), nb, True,
                              lambda l=l: (ho.dbox[l].data_ptr(), nb), rows_level=l)
            if cls_all:
                cls_items.append((cspec, c, clss[l].data_ptr(), lambda l=l: (ho.dcls[l].data_ptr(), ncp)))
            elif cls_fused:
                eng.conv_bias_cls(cspec, c, clss[l].data_ptr(), lambda l=l: (ho.dcls[l].data_ptr(), ncp))
            else:
                eng.conv_bias(cspec, c, clss[l].data_ptr(), ncp, True, lambda l=l: (ho.dcls[l].data_ptr(), ncp))
        if cls_items:
            eng.head_cls_levels(cls_items)
        if box_items:
            eng.head_box_levels(box_items, box_fns)
        if lazy:
            from ...hip import DY_EPI_BIAS, DY_EPI_F32OUT

            def fill():
                assert eng.rec is None, "box logits can only be materialised outside a trace"
                for spec, a, buf, src in lazy:
                    if src is not None:  # the activated input was never written either
                        raw, ps = src
                        eng.call("dy_bn_act_apply", raw.ptr, raw.ld, 0, 0, a.ptr, a.ld, ps.coef.data_ptr(), a.npix, ps.cout, ps.act)
                    eng._conv_raw(spec, a, buf.data_ptr(), nb, DY_EPI_BIAS | DY_EPI_F32OUT, 0, spec.bias)
            ho.fill_box = fill
        return ho

    def _export(self, rt, y):
        if isinstance(y, HeadOut):
            if self.training:
                return y.as_reference_list()
            from ...utils.ops import decode_predictions
            if y.infer is not None:
                return y.infer(), LazyFeats(y)
            # (y, x) like the reference's inference return (head.py:74); the per-level (B, no, H, W) maps are re-formatted from
            # the NHWC head outputs only when somebody indexes them
            return decode_predictions(y), LazyFeats(y)
        return super()._export(rt, y)
