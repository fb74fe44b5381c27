"""The eval forward as a static launch plan (reference get_FPS.py:42-87 times ``model(x)``; engine/predictor.py:150 and
engine/validator.py call the same forward).

The training step got a traced-then-replayed launch list in round 1; the eval forward walked the Python modules on every call and
wrote fp32 head logits only to read them back in a decode launch.  ``InferPlan`` traces ``model.forward_act`` ONCE per input geometry
(B, H, W) -- the stem reads the fp32 NCHW batch directly, every layer's buffers stay resident -- and later calls replay the list
(one hipGraph when the runtime allows it).  Detect's tail (its two final convs per level, DFL decode, sigmoid) is ONE
``dy_head_infer_levels`` launch issued OUTSIDE the recorded list with a freshly allocated output, so the ``y`` a caller receives is
its own tensor, as with the reference, not a static buffer the next forward overwrites.
"""
from __future__ import annotations

import os

import torch

from . import check
from .engine import ImageAct, Recorder

# DY_INFER_PLAN=0: eval ``model(x)`` walks the modules eagerly on every call (import pass, generic final convs, dy_decode_predictions)
INFER_PLAN = os.environ.get("DY_INFER_PLAN", "1") != "0"
NO_INPUT_COPY = os.environ.get("DY_INFER_INPUT_COPY", "0") != "1"
MAX_PLANS = int(os.environ.get("DY_INFER_PLANS", "4"))  # input geometries kept recorded per model (least recently used goes first)


class InferPlan:
    def __init__(self, model, B, H, W, use_graph=None):
        dev = next(model.parameters()).device
        self.model, self.rt = model, model._runtime(dev)
        self.eng = self.rt.eng
        self.shape = (B, 3, H, W)
        self.img = torch.zeros(self.shape, dtype=torch.float32, device=dev)  # the static input the recorded list reads
        from . import GRAPH_SAFE
        self.use_graph = (GRAPH_SAFE and os.environ.get("DY_INFER_GRAPH", "1") != "0") if use_graph is None else bool(use_graph)
        self.rec = self.graph = self.ho = None
        self.keep = []      # this plan's buffers: released with the plan, not parked on the engine for ever
        self.calls = 0
        self.head = 0       # the leading launches that read the image batch: issued per call on the CALLER's tensor (no input copy)

    def _trace(self):
        eng = self.eng
        keep, eng.keep = eng.keep, self.keep
        eng.rec, eng.tape, eng.training, eng.infer_head = Recorder(), None, False, True
        try:
            self.ho = self.model.forward_act(ImageAct(eng, self.img))
        finally:
            self.rec, eng.rec, eng.infer_head, eng.keep = eng.rec, None, False, keep
        # The launches that read the image batch (the stem) stay OUT of the replayed list when they lead it: each call issues them
        # with the caller's own tensor in place of the static input, and a 629 MB device copy per forward (1280x1280 batch 32: 0.25 of
        # 7.6 ms) never happens.  DY_INFER_INPUT_COPY=1, or a list in which something else comes first: copy into ``img`` as before.
        ip = self.img.data_ptr()
        reads = [i for i, (fn, args, _, sid) in enumerate(self.rec.ops) if fn is not None and any(isinstance(v, int) and v == ip for v in args)]
        if NO_INPUT_COPY and reads and reads == list(range(len(reads))) and all(self.rec.ops[i][3] == 0 for i in reads):
            self.head = len(reads)
        if self.use_graph:
            want = self._finish().clone()
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, capture_error_mode="thread_local"):
                eng.replay(self.rec, self.head, None)
            self._head(ip)
            g.replay()
            if not torch.equal(self._finish(), want):  # same safety net as StepPlan._verify_capture: a broken capture must not go unnoticed
                raise RuntimeError("the captured inference graph does not reproduce the traced forward (device work from another host "
                                   "thread during capture, or DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 not in effect: see ultralytics/hip/__init__.py)")
            self.graph = g

    def _head(self, ptr):
        """The leading launches with ``ptr`` where the trace had the static input."""
        eng, ip = self.eng, self.img.data_ptr()
        s = eng.stream
        for fn, args, name, _ in self.rec.ops[:self.head]:
            check(fn(*[ptr if (isinstance(v, int) and v == ip) else v for v in args], s), name)

    def _finish(self):
        """Detect's tail into a fresh tensor (or, for heads the fused kernel does not take, the decode of the logits the list wrote)."""
        if self.ho.infer is not None:
            return self.ho.infer()
        from ..utils.ops import decode_predictions
        return decode_predictions(self.ho)

    def __call__(self, x):
        if tuple(x.shape) != self.shape:
            raise ValueError(f"this plan was recorded for inputs of shape {self.shape}, got {tuple(x.shape)}")
        direct = self.rec is not None and self.head and x.dtype == torch.float32 and x.is_contiguous() and x.device == self.img.device
        if not direct and x.data_ptr() != self.img.data_ptr():
            self.img.copy_(x, non_blocking=True)
        self.rt.ensure_packed()
        self.eng.training = False
        self.calls += 1
        if self.rec is None:
            self._trace()
            return self._finish()
        if self.head:
            self._head(x.data_ptr() if direct else self.img.data_ptr())
        if self.graph is not None:
            self.graph.replay()
        else:
            self.eng.replay(self.rec, self.head, None)
        return self._finish()


def wants_plan(model, x):
    """An eval forward of a detection model on a (B, 3, H, W) device tensor: the case the fused inference path serves."""
    from ..nn.modules import Detect
    return (INFER_PLAN and model.__dict__.get("_capture") is None and torch.is_tensor(x) and x.dim() == 4 and x.shape[1] == 3 and x.is_cuda
            and x.is_floating_point() and isinstance(model.model[-1], Detect))


def forward_eval(model, x):
    """``model(x)`` in eval mode -> (y (B, 4+nc, A) fp32, LazyFeats): the recorded plan of this geometry when there is one, otherwise
    the same launches issued by walking the modules (stem from the image batch, fused Detect tail): bit-identical results either way."""
    from ..nn.modules.head import LazyFeats
    plan = plan_for(model, x)
    with torch.no_grad():
        if plan is not None:
            return plan(x), LazyFeats(plan.ho)
        rt = model._runtime(x.device)
        eng = rt.eng
        eng.training = False
        rt.ensure_packed()
        eng.infer_head = True
        try:
            ho = model.forward_act(ImageAct(eng, x.float().contiguous()))
        finally:
            eng.infer_head = False
        if ho.infer is not None:
            return ho.infer(), LazyFeats(ho)
        from ..utils.ops import decode_predictions
        return decode_predictions(ho), LazyFeats(ho)


def plan_for(model, x):
    """The recorded plan of ``model`` for input ``x`` (B, 3, H, W), or None when this forward should walk the modules: plans are
    made on the SECOND forward of a geometry (a predictor fed differently sized images would otherwise record every one of them)
    and at most MAX_PLANS are kept per model."""
    if not wants_plan(model, x):
        return None
    st = model.__dict__.setdefault("_infer_plans", {"rt": None, "plans": {}, "seen": {}})
    rt = model._runtime(x.device)
    if st["rt"] is not rt:  # parameters were re-created (.to / .half / fuse): every recorded pointer is stale
        st["rt"], st["plans"], st["seen"] = rt, {}, {}
    key = tuple(x.shape)
    plan = st["plans"].get(key)
    if plan is None:
        st["seen"][key] = st["seen"].get(key, 0) + 1
        if st["seen"][key] < 2:
            return None
        if len(st["plans"]) >= MAX_PLANS:
            old = min(st["plans"], key=lambda k: st["plans"][k].last)
            del st["plans"][old]
        plan = st["plans"][key] = InferPlan(model, key[0], key[2], key[3])
    st["tick"] = plan.last = st.get("tick", 0) + 1
    return plan
