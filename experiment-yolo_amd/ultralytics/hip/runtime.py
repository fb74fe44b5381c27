"""Glue between the nn.Module tree (parameter holders with the reference's names) and the HIP engine.

``Runtime.materialize(root)`` moves every parameter of ``root`` into ONE flat fp32 device buffer laid out as
[bias group | decayed weights | norm weights] (the optimizer groups of reference engine/trainer.py:1146-1154), points
``param.data`` / ``param.grad`` at views of it, flattens the floating-point buffers (BN running statistics) likewise,
and builds a ConvSpec per convolution.  Flat storage is what makes the single-launch optimizer/EMA and the single-bucket
RCCL all-reduce possible.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import DY_ACT_NONE, DY_ACT_SILU
from .engine import BN2D_EPS, BN2D_MOM, BN3D_EPS, BN3D_MOM, Act, ConvSpec, Engine, Storage

PAD = 8  # every tensor starts at a multiple of 8 floats inside the flat buffers


def _round(n):
    return (n + PAD - 1) // PAD * PAD


class Runtime:
    def __init__(self, root: nn.Module, device):
        self.root = root
        self.eng = Engine(device)
        self.specs = {}
        self._flatten()
        for m in root.modules():
            if isinstance(m, HipModule):
                m._build_specs(self)
        self._pack_sig = None

    # ---- flat parameter / buffer storage -------------------------------------------------------------------------
    def _flatten(self):
        dev = self.eng.device
        norm_types = tuple(v for k, v in nn.__dict__.items() if "Norm" in k and isinstance(v, type))
        groups = ([], [], [])  # bias, decayed weights, norm weights  (reference build_optimizer order g[2], g[0], g[1])
        seen = set()
        for mname, mod in self.root.named_modules():
            for pname, p in mod.named_parameters(recurse=False):
                if id(p) in seen:
                    continue
                seen.add(id(p))
                full = f"{mname}.{pname}" if mname else pname
                gi = 0 if "bias" in full else (2 if isinstance(mod, norm_types) else 1)
                groups[gi].append((full, p))
        # Six segments [neck + head: bias | decayed | norm][backbone: bias | decayed | norm]: the parameters whose gradients are final
        # FIRST in the backward pass (everything behind the YAML's backbone) sit in front, so each data-parallel gradient bucket is one
        # contiguous slice of the flat gradient buffer that RCCL reduces in place (StepPlan, DY_DP_BUCKETS=2; DDP's reducer works on
        # contiguous buckets too, reference engine/trainer.py:694-695).  A root without a backbone / head split has empty first segments.
        nb = len(self.root.yaml.get("backbone", [])) if isinstance(getattr(self.root, "yaml", None), dict) else None

        def late(name):
            parts = name.split(".")
            return nb is not None and len(parts) > 1 and parts[0] == "model" and parts[1].isdigit() and int(parts[1]) >= nb

        segs = [[(n, p) for n, p in g if late(n)] for g in groups] + [[(n, p) for n, p in g if not late(n)] for g in groups]
        self.param_names = [n for g in segs for n, _ in g]
        self.param_group = {n: k % 3 for k, g in enumerate(segs) for n, _ in g}  # optimizer group of every parameter
        offs, total, bounds = {}, 0, []
        for g in segs:
            for n, p in g:
                offs[n] = total
                total += _round(p.numel())
            bounds.append(total)
        self.n_params_flat, self.seg_bounds = total, bounds[:5]
        self.bucket_split = bounds[2]  # [0, bucket_split): neck + head (the first gradient bucket); the rest: backbone
        groups = segs
        self.flat_p = torch.zeros(total, dtype=torch.float32, device=dev)
        self._n_params_for_gb = total  # flat_g is created with the buffers' staging tail once their size is known (below)
        self.frozen = torch.zeros(total, dtype=torch.uint8, device=dev)
        self.param_off = offs
        # floating-point buffers (running statistics)
        fb = []
        for mname, mod in self.root.named_modules():
            for bname, b in mod.named_buffers(recurse=False):
                if b is not None and b.dtype.is_floating_point:
                    fb.append((mod, bname, b))
        tot = sum(_round(b.numel()) for _, _, b in fb)
        # ONE exchange buffer [gradients | staging copy of the float buffers]: the data-parallel step all-reduces it whole, so the
        # DDP-style "rank 0's BN statistics win" rides on the gradient all-reduce (rank 0 stages its buffers, the others zeros)
        self.flat_gb = torch.zeros(total + max(tot, PAD), dtype=torch.float32, device=dev)
        self.flat_g = self.flat_gb[:total]
        self.gviews = {}
        for g in groups:
            for n, p in g:
                o, k = offs[n], p.numel()
                view = self.flat_p[o:o + k].view(p.shape)
                view.copy_(p.data.to(dev, torch.float32))
                p.data = view
                gv = self.flat_g[o:o + k].view(p.shape)
                p.grad = gv
                self.gviews[id(p)] = gv
                if not p.requires_grad:
                    self.frozen[o:o + _round(k)] = 1
                else:
                    self.frozen[o + k:o + _round(k)] = 1  # padding never moves
        self.flat_b = torch.zeros(max(tot, PAD), dtype=torch.float32, device=dev)
        o = 0
        for mod, bname, b in fb:
            k = b.numel()
            view = self.flat_b[o:o + k].view(b.shape)
            view.copy_(b.to(dev, torch.float32))
            mod._buffers[bname] = view
            o += _round(k)
        self.n_buffers_flat = max(tot, PAD)
        # integer buffers just move
        for mod in self.root.modules():
            for bname, b in list(mod._buffers.items()):
                if b is not None and not b.dtype.is_floating_point:
                    mod._buffers[bname] = b.to(dev)
            if isinstance(mod, HipModule):
                mod.__dict__["rt"] = self

    def refresh_frozen(self):
        """Re-read ``requires_grad`` flags (BaseTrainer freezes '.dfl' after construction, engine/trainer.py:670-683)."""
        self.frozen.zero_()
        for n, p in self.root.named_parameters():
            o, k = self.param_off[n], p.numel()
            if not p.requires_grad:
                self.frozen[o:o + _round(k)] = 1
            else:
                self.frozen[o + k:o + _round(k)] = 1

    # ---- conv specs -----------------------------------------------------------------------------------------------
    def _bn_dict(self, bn):
        return dict(weight=bn.weight.data, bias=bn.bias.data, running_mean=bn.running_mean, running_var=bn.running_var,
                    nbt=bn.num_batches_tracked)

    def make_spec(self, key, conv: nn.Module, bn, act, ks, stride, eps=BN2D_EPS, mom=BN2D_MOM, name="", ld=None):
        sp = self.specs.get(key)
        if sp is not None and sp.weight.data_ptr() == conv.weight.data.data_ptr() and (sp.bn is None) == (bn is None):
            return sp
        w = conv.weight.data
        w2 = w if ld is not None else (w.reshape(w.shape[0], w.shape[1], ks, ks) if w.dim() != 4 or w.shape[2] != ks else w)
        sp = ConvSpec(name, w2, None if conv.bias is None else conv.bias.data, None if bn is None else self._bn_dict(bn), ks,
                      stride, act, eps, mom)
        sp.ld = ld
        sp.gweight = self.gviews.get(id(conv.weight))
        sp.gbias = None if conv.bias is None else self.gviews.get(id(conv.bias))
        if bn is not None:
            sp.gbn_w, sp.gbn_b = self.gviews.get(id(bn.weight)), self.gviews.get(id(bn.bias))
        self.eng.prepare_conv(sp)
        self.specs[key] = sp
        sp.packed = False
        return sp

    def spec(self, m):
        """ConvSpec of a ``Conv`` module (fused or not)."""
        return self.specs[id(m)]

    def ldconv(self, m, x, out=None):
        sp_p, sp_c = self.specs[(id(m), "p_conv")], self.specs[(id(m), "conv", x.C)]
        return self.eng.ldconv(sp_p, sp_c, m._pn_i32, m.num_param, m.stride, x, out)

    def pack_all(self, transposed=True):
        """Refresh every MFMA weight pack from the fp32 masters in ONE launch (recorded at the start of each step)."""
        import ctypes as C
        if not self.specs:
            return
        key = (transposed, tuple(sp.weight.data_ptr() for sp in self.specs.values()))
        cached = getattr(self, "_pack_tbl", None)
        if cached is None or cached[0] != key:
            L = self.eng.L
            sz = L.dy_pack_desc_bytes()
            n = len(self.specs) * (2 if transposed else 1)
            host = (C.c_char * (sz * n))()
            blocks, i = 0, 0
            for sp in self.specs.values():
                for tr in ((0, 1) if transposed else (0,)):
                    ld = sp.ld or (0, 0, sp.weight.shape[1])
                    nb = L.dy_pack_desc_fill(C.byref(host, i * sz), sp.weight.data_ptr(), 0, (sp.wpack_t if tr else sp.wpack).data_ptr(),
                                             sp.cout, ld[2] if sp.ld else sp.weight.shape[1], sp.ks, sp.stride, tr, ld[0], ld[1], blocks)
                    if nb < 0:
                        raise RuntimeError(f"dy_pack_desc_fill failed for {sp.name}")
                    blocks += nb
                    i += 1
            dev = torch.frombuffer(bytearray(host), dtype=torch.uint8).to(self.eng.device)
            self.eng.keep.append(dev)
            self._pack_tbl = cached = (key, dev, n, blocks)
        self.eng.call("dy_pack_weights_batched", cached[1].data_ptr(), cached[2], cached[3])

    def ensure_packed(self):
        """Eager (inference) path: re-pack when any master weight changed (torch version counters + explicit marks)."""
        sig = (self._dirty_mark, tuple(sp.weight._version for sp in self.specs.values()))
        if sig != self._pack_sig:
            self.pack_all(transposed=False)
            self._pack_sig = sig

    _dirty_mark = 0

    def mark_dirty(self):
        self._dirty_mark += 1

    # ---- tensor <-> Act ------------------------------------------------------------------------------------------
    def to_act(self, x: torch.Tensor) -> Act:
        """Public-API tensor (N,C,H,W) -> Act.  C not a multiple of 8 (the RGB image) is zero-padded."""
        eng = self.eng
        if x.dim() != 4:
            raise ValueError(f"expected a (N,C,H,W) tensor, got {tuple(x.shape)}")
        x = x.to(eng.device)
        N, Cc, H, W = x.shape
        if Cc % 8:
            return eng.import_image(x.float().contiguous(), (Cc + 7) // 8 * 8)
        st = Storage(eng, N, H, W, Cc)
        st.buf.copy_(x.permute(0, 2, 3, 1))
        return st.act()

    @staticmethod
    def to_tensor(a: Act) -> torch.Tensor:
        """Act -> logical (N,C,H,W) fp16 tensor sharing memory with the NHWC buffer (channels-last strides)."""
        return a.st.buf[..., a.c0:a.c0 + a.C].permute(0, 3, 1, 2)


class HipModule(nn.Module):
    """Base of every hot-path module: ``forward`` accepts public tensors or engine Acts and routes to ``forward_act``."""

    rt: Runtime | None = None

    def _build_specs(self, rt):
        """Register this module's ConvSpecs with the runtime (overridden by modules that own convolutions)."""

    def _runtime(self, device):
        rt = self.__dict__.get("rt")
        if rt is None or rt.eng.device != torch.device(device):
            rt = Runtime(self, device)  # standalone use of a sub-module: it becomes its own root
        return rt

    def _apply(self, fn, *a, **k):
        out = super()._apply(fn, *a, **k)
        for m in self.modules():  # parameters were re-created: flat views are stale
            if isinstance(m, HipModule):
                m.__dict__.pop("rt", None)
        return out

    def forward(self, x, *args, **kwargs):
        first = x[0] if isinstance(x, (list, tuple)) else x
        if isinstance(first, Act):
            return self.forward_act(x, *args, **kwargs)
        dev = first.device
        if dev.type != "cuda":
            raise RuntimeError(f"{type(self).__name__}: the HIP hot path runs on the GPU only (got a {dev.type} tensor); "
                               "there is deliberately no CPU fallback")
        rt = self._runtime(dev)
        rt.eng.training = self.training
        xs = [rt.to_act(t) for t in x] if isinstance(x, (list, tuple)) else rt.to_act(x)
        with torch.no_grad():
            rt.ensure_packed()
            y = self.forward_act(xs, *args, **kwargs)
        return self._export(rt, y)

    def _export(self, rt, y):
        from .engine import SegAct, UpAct
        if isinstance(y, (SegAct, UpAct)):
            y = rt.eng.dense(y)
        if isinstance(y, Act):
            return rt.to_tensor(y)
        if isinstance(y, (list, tuple)):
            return type(y)(self._export(rt, t) for t in y)
        return y
