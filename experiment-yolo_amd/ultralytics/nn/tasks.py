"""Model graph: YAML -> module list -> DetectionModel (drop-in for the detection part of reference nn/tasks.py).

Differences from the reference that are deliberate:
  * strides are derived statically from the graph instead of a zeros(2,3,640,640) probe forward (tasks.py:309-317), so a
    model can be built on a host without GPU;
  * the forward runs on the HIP engine (NHWC fp16 Acts); ``model(batch_dict)`` returns ``(loss*B, loss_items)`` exactly as
    ``BaseModel.forward`` does (tasks.py:63-65) with both tensors living on the device and no host synchronisation.
"""
from __future__ import annotations

import contextlib
import math
import re
from copy import deepcopy
from pathlib import Path

import torch
import torch.nn as nn
import yaml

from ..hip.engine import Act
from ..hip.runtime import HipModule, Runtime
from .extra_modules.block import Add, ScalSeq, Zoom_cat
from .modules import SPPF, C2f, Concat, Conv, Detect, LDConv

CFG_MODELS = Path(__file__).resolve().parent.parent / "cfg" / "models"


def make_divisible(x, divisor=8):
    return int(math.ceil(x / divisor) * divisor)


class Upsample(HipModule):
    """nn.Upsample(None, 2, 'nearest') of the model YAMLs, on the HIP engine."""

    def __init__(self, size=None, scale_factor=None, mode="nearest"):
        super().__init__()
        if size is not None or int(scale_factor) != 2 or mode != "nearest":
            raise NotImplementedError("only nn.Upsample(None, 2, 'nearest') is on the hot path")
        self.size, self.scale_factor, self.mode = size, float(scale_factor), mode

    def forward_act(self, x, out=None):
        return self.rt.eng.upsample2x(x, out)


_MODULES = {"Conv": Conv, "LDConv": LDConv, "C2f": C2f, "SPPF": SPPF, "Concat": Concat, "nn.Upsample": Upsample,
            "ScalSeq": ScalSeq, "Add": Add, "Zoom_cat": Zoom_cat, "Detect": Detect}


def guess_model_scale(model_path):
    """Scale letter from 'yolov8[nslmx]...' (reference tasks.py:1083-1099)."""
    with contextlib.suppress(AttributeError):
        return re.search(r"yolov\d+([nslmx])", Path(model_path).stem).group(1)
    return ""


def yaml_model_load(path):
    """'yolov8n-X.yaml' is served by 'yolov8-X.yaml' with scale 'n' (reference tasks.py:1065-1080); bare names resolve
    under ultralytics/cfg/models."""
    path = Path(path)
    unified = Path(re.sub(r"(\d+)([nslmx])(.+)?$", r"\1\3", str(path)))
    cands = [unified, path, CFG_MODELS / unified.name, CFG_MODELS / path.name]
    src = next((c for c in cands if c.is_file()), None)
    if src is None:
        raise FileNotFoundError(f"model YAML '{path}' not found (searched {[str(c) for c in cands]})")
    d = yaml.safe_load(src.read_text(errors="ignore"))
    d["scale"] = guess_model_scale(path)
    d["yaml_file"] = str(path)
    return d


def parse_model(d, ch, verbose=True):
    """YAML dict -> (nn.Sequential, savelist, per-layer down-sampling) for the hot-path module names
    (reference tasks.py:780-1062, branches :825-864, :905-911, :1001-1008)."""
    nc, scales = d.get("nc"), d.get("scales")
    depth, width, max_channels = d.get("depth_multiple", 1.0), d.get("width_multiple", 1.0), float("inf")
    if scales:
        scale = d.get("scale") or tuple(scales.keys())[0]
        depth, width, max_channels = scales[scale]
    chs, down = [ch], [1.0]
    layers, save = [], []
    for i, (f, n, mname, args) in enumerate(d["backbone"] + d["head"]):
        if mname not in _MODULES:
            raise NotImplementedError(f"module '{mname}' is outside the DEAL-YOLO hot path (supported: {sorted(_MODULES)})")
        m = _MODULES[mname]
        args = [nc if a == "nc" else (None if a == "None" else a) for a in args]
        n_ = n = max(round(n * depth), 1) if n > 1 else n
        fl = [f] if isinstance(f, int) else list(f)
        if m in (Conv, LDConv, C2f, SPPF):
            c1, c2 = chs[f], args[0]
            if c2 != nc:
                c2 = make_divisible(min(c2, max_channels) * width, 8)
            args = [c1, c2, *args[1:]]
            if m is C2f:
                args.insert(2, n)
                n = 1
            s = args[3] if (m in (Conv, LDConv) and len(args) > 3) else 1
            ds = down[f] * s
        elif m is Upsample:
            c2, ds = chs[f], down[f] / int(args[1])
        elif m is Concat:
            c2, ds = sum(chs[x] for x in fl), down[fl[0]]
        elif m is Zoom_cat:
            c2, ds = sum(chs[x] for x in fl), down[fl[1]]
        elif m is Add:
            c2, ds = chs[fl[-1]], down[fl[-1]]
        elif m is ScalSeq:
            c1 = [chs[x] for x in fl]
            c2 = make_divisible(args[0] * width, 8)
            args, ds = [c1, c2], down[fl[0]]
        elif m is Detect:
            args.append([chs[x] for x in fl])
            c2, ds = sum(args[1]) if False else nc + 64, down[fl[0]]
        m_ = nn.Sequential(*(m(*args) for _ in range(n))) if n > 1 else m(*args)
        t = f"{m.__module__}.{m.__name__}" if m is not Upsample else "torch.nn.modules.upsampling.Upsample"
        m_.np = sum(x.numel() for x in m_.parameters())
        m_.i, m_.f, m_.type = i, f, t
        if m is Detect:
            m_.stride = torch.tensor([down[x] for x in fl], dtype=torch.float32)
        save.extend(x % i for x in fl if x != -1)
        layers.append(m_)
        if i == 0:
            chs, down = [], []
        chs.append(c2)
        down.append(ds)
        _ = n_
    return nn.Sequential(*layers), sorted(save)


def initialize_weights(model):
    """BN eps/momentum and in-place activations (reference utils/torch_utils.py:342-352; BatchNorm3d is untouched)."""
    for m in model.modules():
        t = type(m)
        if t is nn.BatchNorm2d:
            m.eps, m.momentum = 1e-3, 0.03
        elif t in (nn.Hardswish, nn.LeakyReLU, nn.ReLU, nn.ReLU6, nn.SiLU):
            m.inplace = True


class BaseModel(HipModule):
    """forward / predict / fuse / loss protocol of reference tasks.py:52-270."""

    def forward(self, x, *args, **kwargs):
        if isinstance(x, dict):
            return self.loss(x, *args, **kwargs)
        return self.predict(x, *args, **kwargs)

    def predict(self, x, profile=False, visualize=False, augment=False, embed=None):
        """Reference tasks.py:67-83.  In eval mode a detection model's forward of a (B, 3, H, W) device tensor goes through the
        inference plan of its geometry (hip/infer.py: recorded launch list / hipGraph, stem from the image batch, fused Detect tail)."""
        if not self.training:
            from ..hip.infer import forward_eval, wants_plan
            if wants_plan(self, x):
                return forward_eval(self, x)
        return HipModule.forward(self, x)

    def _concat_plan(self):
        """layer index -> (concat layer index, channel offset, total channels): producers that can write straight into a
        Concat's buffer (each producer is claimed by its first Concat consumer)."""
        plan, chans = {}, []
        for m in self.model:
            chans.append(None)
        for m in self.model:
            if isinstance(m, Concat):
                fl = [m.i + j if j < 0 else j for j in m.f]
                if len(set(fl)) != len(fl) or any(j in plan for j in fl):
                    continue
                ok = all(isinstance(self.model[j], (Conv, LDConv, C2f, SPPF, Upsample, Add, ScalSeq)) for j in fl)
                if ok:
                    for pos, j in enumerate(fl):
                        plan[j] = (m.i, pos)
        return plan

    def forward_act(self, x, out=None):
        """BaseModel._predict_once (reference tasks.py:85-126) over engine Acts."""
        eng = self.rt.eng
        from ..hip.engine import PLANAR, SegAct, UpAct
        # (PLANAR: a Concat's inputs stay where their producers put them -- the 1x1 conv behind it reads them through a segment table)
        plan = {} if PLANAR else self._concat_plan()
        ys, cats = [], {}
        from ..hip.engine import ImageAct
        if isinstance(x, ImageAct) and not (type(self.model[0]) is Conv and self.model[0].f == -1 and 0 not in plan):
            x = x.materialize()  # only a plain Conv stem reads the image batch directly (csrc/stem.hip)
        n_backbone = len(self.yaml.get("backbone", [])) if isinstance(getattr(self, "yaml", None), dict) else 0
        from ..hip.engine import SCALSEQ_ADD
        folded = None
        for m in self.model:
            if m.i == n_backbone and eng.tape is not None:
                eng.tape_mark = len(eng.tape)  # backward closures from here on belong to the neck + head (StepPlan's gradient buckets)
            if folded is not None and m.i == folded:
                folded = None  # this Add was computed by the ScalSeq in front of it (x is the sum already)
                ys.append(x if m.i in self.save else None)
                continue
            if m.f != -1:
                x = ys[m.f] if isinstance(m.f, int) else [x if j == -1 else ys[j] for j in m.f]
            dst = None
            if m.i in plan:
                ci, pos = plan[m.i]
                cat_m = self.model[ci]
                fl = [cat_m.i + j if j < 0 else j for j in cat_m.f]
                src0 = x[0] if isinstance(x, (list, tuple)) else x
                if isinstance(m, Upsample):
                    oh, ow = 2 * src0.H, 2 * src0.W
                elif hasattr(m, "out_hw"):
                    oh, ow = m.out_hw(src0.H, src0.W)
                else:
                    oh, ow = src0.H, src0.W
                if ci not in cats:
                    widths = [self._cout[j] for j in fl]
                    cats[ci] = (eng.new_storage(src0.N, oh, ow, sum(widths)), widths)
                st, widths = cats[ci]
                dst = st.act(sum(widths[:pos]), widths[pos])
            nxt = self.model[m.i + 1] if m.i + 1 < len(self.model) else None
            if (SCALSEQ_ADD and self.__dict__.get("_capture") is None  # (a per-layer capture wants ScalSeq's own output to exist)
                    and isinstance(m, ScalSeq) and dst is None and isinstance(nxt, Add) and nxt.i not in plan
                    and m.i not in self.save and isinstance(nxt.f, (list, tuple)) and len(nxt.f) == 2
                    and sum(1 for j in nxt.f if j == -1 or j == m.i) == 1):
                # ScalSeq -> Add([other, -1]) (reference yolov8-ASF*.yaml): the sum rides on ScalSeq's tail kernel
                other = [j for j in nxt.f if not (j == -1 or j == m.i)][0]
                x = m.forward_act(x, res=ys[other])
                folded = nxt.i
                ys.append(None)
                continue
            x = m.forward_act(x, dst) if dst is not None else m.forward_act(x)
            ys.append(x if m.i in self.save else None)
            if self.__dict__.get("_capture") is not None:  # tests: per-layer outputs (engine Acts) of this forward
                self._capture.append(eng.snapshot(x))
        return x

    def _export(self, rt, y):
        from .modules.head import HeadOut
        if isinstance(y, HeadOut):
            return self.model[-1]._export(rt, y)
        return super()._export(rt, y)

    def fuse(self, verbose=True):
        """Fold BatchNorm into every ``Conv`` (reference tasks.py:168-195, utils/torch_utils.py:171-198).  LDConv's inner BN
        and ScalSeq's BatchNorm3d are left alone, as in the reference."""
        if not self.is_fused():
            for m in self.modules():
                if isinstance(m, Conv) and hasattr(m, "bn"):
                    conv, bn = m.conv, m.bn
                    fused = nn.Conv2d(conv.in_channels, conv.out_channels, conv.kernel_size, conv.stride, conv.padding,
                                      bias=True).requires_grad_(False).to(conv.weight.device)
                    scale = bn.weight.detach().float() / torch.sqrt(bn.eps + bn.running_var.detach().float())
                    fused.weight.copy_(conv.weight.detach().float() * scale.view(-1, 1, 1, 1))
                    fused.bias.copy_(bn.bias.detach().float() - bn.running_mean.detach().float() * scale)
                    m.conv = fused
                    delattr(m, "bn")
            for m in self.modules():
                if isinstance(m, HipModule):
                    m.__dict__.pop("rt", None)
        return self

    def is_fused(self, thresh=10):
        bn = tuple(v for k, v in nn.__dict__.items() if "Norm" in k and isinstance(v, type))
        return sum(isinstance(v, bn) for v in self.modules()) < thresh

    def info(self, detailed=False, verbose=True, imgsz=640):
        n_p = sum(x.numel() for x in self.parameters())
        n_l = len(list(self.modules()))
        if verbose:
            print(f"{type(self).__name__} summary: {n_l} layers, {n_p} parameters")
        return n_l, n_p

    def load(self, weights, verbose=True):
        model = weights["model"] if isinstance(weights, dict) else weights
        csd = model.float().state_dict() if hasattr(model, "state_dict") else model
        own = self.state_dict()
        csd = {k: v for k, v in csd.items() if k in own and own[k].shape == v.shape}
        self.load_state_dict(csd, strict=False)
        return len(csd)

    def loss(self, batch, preds=None):
        """Reference tasks.py:256-268: lazily build the criterion, ``preds = self.forward(batch["img"]) if preds is None else
        preds``, ``return self.criterion(preds, batch)`` -> (loss.sum()*B, loss_items[box, cls, dfl]).  The forward keeps the raw
        head outputs on the engine (no NCHW re-formatting).

        In training mode with autograd enabled the returned loss carries a ``grad_fn`` -- ONE custom ``torch.autograd.Function``
        whose backward replays the recorded backward launch list and ACCUMULATES into the ``.grad`` views of the flat gradient
        buffer -- so the reference's loop shape trains (engine/trainer.py:802-815):
        ``loss, items = model(batch); scaler.scale(loss).backward(); scaler.step(optimizer)`` with any ``torch.optim`` optimizer
        over ``model.parameters()``."""
        if not hasattr(self, "criterion"):
            self.criterion = self.init_criterion()
        if preds is None and self.training and torch.is_grad_enabled():
            return self._loss_with_grad(batch)
        if preds is None:
            img = batch["img"]
            dev = next(self.parameters()).device
            if dev.type != "cuda":
                raise RuntimeError("the HIP hot path runs on the GPU only: move the model with .cuda() first (no CPU fallback)")
            rt = self._runtime(dev)
            rt.eng.training = self.training
            with torch.no_grad():
                rt.ensure_packed()
                if img.dtype == torch.uint8:
                    img = img.float() / 255
                preds = self.forward_act(rt.to_act(img.to(dev)))
        return self.criterion(preds, batch)

    def _loss_with_grad(self, batch):
        from ..hip.train import StepPlan
        img = batch["img"]
        dev = next(self.parameters()).device
        if dev.type != "cuda":
            raise RuntimeError("the HIP hot path runs on the GPU only: move the model with .cuda() first (no CPU fallback)")
        u8 = img.dtype == torch.uint8 and img.shape[-1] == 3 and img.shape[1] != 3
        B, (H, W) = img.shape[0], (img.shape[1:3] if u8 else img.shape[2:4])
        nmax = self.criterion.capacity_for(batch, B)
        plans = self.__dict__.setdefault("_ag_plans", {})
        key = (B, int(H), int(W), u8, dev)
        plan = plans.get(key)
        if plan is None or plan.nmax < nmax or plan.rt is not self.__dict__.get("rt"):
            # one recorded launch list per input geometry; fp16 backward at an internal loss scale that follows GradScaler's policy
            first = next((p for p in plans.values() if p.rt is self.__dict__.get("rt")), None)  # optimizer-side state is shared
            plan = plans[key] = StepPlan(self, B, (int(H), int(W)), nmax=max(16, 2 * nmax), use_graph=False, init_scale=1024.0, share=first)
        if img.dtype == torch.uint8 and not u8:
            batch = dict(batch, img=img.float() / 255)
        s = plan.forward_only({k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in batch.items()})
        anchor = self.__dict__.get("_ag_anchor")
        if anchor is None or anchor.device != dev:
            anchor = self.__dict__["_ag_anchor"] = torch.zeros(1, device=dev, requires_grad=True)
        return _PlanLoss.apply(anchor, plan, s[8].clone()), s[5:8].clone()

    def init_criterion(self):
        raise NotImplementedError


class _PlanLoss(torch.autograd.Function):
    """grad_fn of the loss ``BaseModel.loss`` returns in training mode: its backward IS the recorded backward launch list of the
    step plan whose forward produced the value (reference: autograd through v8DetectionLoss and every module, engine/trainer.py:810)."""

    @staticmethod
    def forward(ctx, anchor, plan, value):
        ctx.plan = plan
        ctx.serial = plan.fwd_serial = getattr(plan, "fwd_serial", 0) + 1
        return value.clone()

    @staticmethod
    def backward(ctx, grad_out):
        plan = ctx.plan
        if ctx.serial != plan.fwd_serial:
            raise RuntimeError("backward() of a loss whose forward activations have been overwritten by a later model(batch) call "
                               "(one forward / backward pair at a time per input geometry)")
        plan.backward_accumulate(grad_out)
        return None, None, None


class DetectionModel(BaseModel):
    """YOLOv8-style detection model built from a YAML (reference tasks.py:275-378)."""

    def __init__(self, cfg="yolov8n.yaml", ch=3, nc=None, verbose=True):
        super().__init__()
        self.yaml = cfg if isinstance(cfg, dict) else yaml_model_load(cfg)
        ch = self.yaml["ch"] = self.yaml.get("ch", ch)
        if nc and nc != self.yaml["nc"]:
            self.yaml["nc"] = nc
        self.model, self.save = parse_model(deepcopy(self.yaml), ch=ch, verbose=verbose)
        self.names = {i: f"{i}" for i in range(self.yaml["nc"])}
        self.inplace = self.yaml.get("inplace", True)
        self._cout = self._layer_channels(ch)
        m = self.model[-1]
        if isinstance(m, Detect):
            m.inplace = self.inplace
            self.stride = m.stride
            self._stride_probe_side_effects()
            m.bias_init()
        else:
            self.stride = torch.Tensor([32])
        initialize_weights(self)
        if verbose:
            self.info()

    def _stride_probe_side_effects(self):
        """The reference derives the strides from a TRAIN-mode forward of ``zeros(1, ch, 256, 256)`` (tasks.py:309-317) -- before
        ``initialize_weights`` sets the BatchNorm momentum, so with nn.BatchNorm's default 0.1.  Every activation of that
        forward is exactly zero (bias-free convs, BN beta = 0, SiLU(0) = 0; LDConv samples zeros whatever its offsets; ScalSeq's
        Conv3d adds its bias, a per-channel constant that its BatchNorm3d removes again), so its only lasting effect is on the
        BN buffers: running_var = 0.9 * 1 + 0.1 * 0, num_batches_tracked = 1, and running_mean = 0.1 * conv3d.bias in ScalSeq.
        Strides are static here; this reproduces the buffers a freshly built reference model starts training from (checked
        against the reference in tests/test_host_logic.py; the reference's rounding noise of ~1e-8 in a few means is not)."""
        for mod in self.modules():
            if isinstance(mod, (nn.BatchNorm2d, nn.BatchNorm3d)):
                mod.running_var.fill_(0.9)
                mod.num_batches_tracked.fill_(1)
            if isinstance(mod, ScalSeq):
                mod.bn.running_mean.copy_(0.1 * mod.conv3d.bias.detach())

    def _layer_channels(self, ch):
        out = []
        for m in self.model:
            if isinstance(m, Conv):
                out.append(m.conv.out_channels)
            elif isinstance(m, LDConv):
                out.append(m.conv[0].out_channels)
            elif isinstance(m, (C2f, SPPF)):
                out.append(m.cv2.conv.out_channels)
            elif isinstance(m, Upsample):
                out.append(out[m.i + m.f if m.f < 0 else m.f])
            elif isinstance(m, (Concat, Zoom_cat)):
                out.append(sum(out[m.i + j if j < 0 else j] for j in m.f))
            elif isinstance(m, Add):
                out.append(out[m.i + m.f[-1] if m.f[-1] < 0 else m.f[-1]])
            elif isinstance(m, ScalSeq):
                out.append(m.conv3d.out_channels)
            else:
                out.append(getattr(m, "no", 0))
        return out

    def _apply(self, fn, *a, **k):
        out = super()._apply(fn, *a, **k)
        m = self.model[-1]
        if isinstance(m, Detect):
            m.stride = fn(m.stride)
            self.stride = m.stride
        return out

    def init_criterion(self):
        from ..utils.loss import v8DetectionLoss
        return v8DetectionLoss(self)


class _Opaque:
    """Stand-in for a pickled class this package does not define (loss objects, namespaces, callbacks ... hanging off a
    reference checkpoint): takes any state and is never executed."""

    def __init__(self, *a, **k):
        pass

    def __setstate__(self, state):
        if isinstance(state, dict):
            self.__dict__.update(state)


def torch_safe_load(weight):
    """torch.load for checkpoints in the reference's on-disk format (reference tasks.py:649-703): whole-module pickles
    whose class paths (``ultralytics.nn.modules.conv.Conv`` ...) resolve to this package's classes of the same name;
    classes under ``ultralytics.*`` that do not exist here unpickle as inert ``_Opaque`` objects.  Returns (ckpt, file)."""
    import pickle
    import types

    class _Unpickler(pickle.Unpickler):
        def find_class(self, module, name):
            try:
                return super().find_class(module, name)
            except (ImportError, AttributeError):
                if module.split(".")[0] == "ultralytics":
                    return type(name, (_Opaque,), {"__module__": module})
                raise

    shim = types.SimpleNamespace(Unpickler=_Unpickler, load=pickle.load, loads=pickle.loads, __name__="pickle")
    ckpt = torch.load(weight, map_location="cpu", weights_only=False, pickle_module=shim)
    if not isinstance(ckpt, dict):
        ckpt = {"model": ckpt}
    return ckpt, weight


def _rebuild(src, ckpt):
    """A usable DetectionModel from what a checkpoint carries: a pickled module (reference or ours: only its ``yaml`` and
    ``state_dict()`` are trusted -- the unpickled object itself never ran this package's constructors) or a state dict
    next to a ``yaml`` entry (this package's trainer)."""
    if isinstance(src, nn.Module):
        cfg, sd = getattr(src, "yaml", None), src.state_dict()
        names, args = getattr(src, "names", None), getattr(src, "args", None)
    else:
        cfg, sd, names, args = ckpt.get("yaml"), src, None, None
    if not isinstance(cfg, dict):
        raise TypeError("checkpoint carries no model YAML: cannot rebuild the network")
    model = DetectionModel(deepcopy(cfg), ch=cfg.get("ch", 3), verbose=False)
    model.load_state_dict({k: v.float() for k, v in sd.items()}, strict=True)
    if isinstance(names, dict) and len(names) == len(model.names):
        model.names = names
    model.args = ckpt.get("train_args", args if isinstance(args, dict) else {})
    model.pt_path = None
    return model


def reference_module_copy(model):
    """A detached fp16 copy of ``model`` shaped like the object the reference pickles (engine/trainer.py:907-908
    ``deepcopy(model).half()``): same class paths and attribute names, no engine state, ``nn.Upsample`` for the up-sampling
    layers, the ``anchors`` / ``strides`` / ``args`` / ``warehouse_manager`` attributes the reference's classes expect.
    The reference unpickles it into its own classes and runs it; LDConv's backward hook is not part of a pickle and the
    reference re-registers it only when it builds the module itself."""
    stash = {}
    for m in model.modules():
        stash[m] = {k: m.__dict__.pop(k) for k in ("rt", "_pn_i32", "_infer_plans", "_ag_plans", "_ag_anchor") if k in m.__dict__}
    try:
        cp = deepcopy(model)
    finally:
        for m, d in stash.items():
            m.__dict__.update(d)
    cp = cp.cpu().half()
    cp.__dict__.pop("_cout", None)
    cp.__dict__.setdefault("warehouse_manager", None)
    if not isinstance(cp.__dict__.get("args"), dict):
        cp.__dict__["args"] = {}
    seq = cp.model
    for name, m in list(seq._modules.items()):
        if isinstance(m, Upsample):
            up = nn.Upsample(m.size, m.scale_factor, m.mode)
            for k in ("i", "f", "type", "np"):
                if hasattr(m, k):
                    setattr(up, k, getattr(m, k))
            seq._modules[name] = up
        elif isinstance(m, Detect):
            m.__dict__.setdefault("shape", None)
            m.__dict__["anchors"], m.__dict__["strides"] = torch.empty(0), torch.empty(0)
    return cp


def save_reference_format(path, model, ema=None, updates=0, epoch=-1, best_fitness=None, train_args=None):
    """Write a checkpoint with the reference's keys and object layout (engine/trainer.py:898-923); ``attempt_load_weights``
    of either side reads it back."""
    import datetime
    ckpt = {"epoch": epoch, "best_fitness": best_fitness, "model": reference_module_copy(model),
            "ema": reference_module_copy(ema) if ema is not None else None, "updates": updates, "optimizer": None,
            "train_args": dict(train_args or {}), "date": datetime.datetime.now().isoformat(), "version": "8.1.9"}
    torch.save(ckpt, path)
    return path


def attempt_load_weights(weights, device=None, inplace=True, fuse=False):
    """Load a checkpoint (reference tasks.py:706-777 attempt_load_weights / attempt_load_one_weight): the reference's
    whole-module ``.pt`` files (fp16 ``model`` / ``ema``) as well as this package's state-dict checkpoints."""
    ckpt, _ = torch_safe_load(weights)
    src = ckpt.get("ema") if ckpt.get("ema") is not None else ckpt["model"]
    model = _rebuild(src, ckpt)
    if device is not None:
        model = model.to(device)
    if fuse and hasattr(model, "fuse"):
        model = model.fuse()
    return model.eval()
