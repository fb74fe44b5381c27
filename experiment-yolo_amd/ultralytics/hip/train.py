"""One training step of DEAL-YOLO as a static launch plan (reference hot loop engine/trainer.py:780-815, 949-957).

The first call traces forward + loss + backward (+ optimizer) through the Python modules while every launch is
recorded; later calls replay the recorded lists -- optionally captured into hipGraphs -- so steady-state steps touch no
module code, no allocator and no host synchronisation.  Gradients of all ranks are summed with ONE RCCL all-reduce over
the flat fp32 gradient buffer (the reference multiplies the loss by world_size and lets DDP average: same result).
"""
from __future__ import annotations

import ctypes as C
import math
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

from . import check, lib
from .dist import all_reduce_flat
from .engine import Recorder


class StepPlan:
    def __init__(self, model, batch_size, imgsz, nmax=16, optimizer="SGD", hyp=None, world_size=1, use_graph=False,
                 init_scale=65536.0, side_wgrad=None, dynamic_scale=True, share=None, arena=None, input_act=False):
        """``share``: another StepPlan on the SAME model whose optimizer state this one uses (momentum / Adam moments, EMA, loss
        scale and step counters, hyper-parameters): a plan records ONE batch size, so the ragged last batch of an epoch gets its
        own forward/backward launch list while accumulate / all_reduce / optimizer_step stay with the main plan.
        ``arena``: an ``engine.Arena`` this plan lays its step-local buffers over (shared by the plans of a multi-scale run).
        ``input_act``: the recorded list starts at the stem's fp16 NHWC input ``self.x_in`` (B, H, W, 8; channels 3..7 stay zero),
        filled by the caller before every ``forward_backward`` -- the multi-scale trainer writes the re-interpolated batch there."""
        self.model = model
        self.arena = arena
        # weight gradients on a second stream beside the input-gradient chain (env DY_SIDE_WGRAD=0/1 overrides the default)
        self.side_wgrad = bool(int(os.environ.get("DY_SIDE_WGRAD", "0"))) if side_wgrad is None else bool(side_wgrad)
        dev = next(model.parameters()).device
        self.rt = model._runtime(dev)
        self.eng = self.rt.eng
        self.B, self.imgsz, self.nmax = batch_size, (imgsz, imgsz) if isinstance(imgsz, int) else tuple(imgsz), nmax
        self.world_size = world_size
        if use_graph:
            from . import GRAPH_SAFE
            if not GRAPH_SAFE and not os.environ.get("DY_ALLOW_UNSAFE_GRAPHS"):
                raise RuntimeError("hipGraph replay is not safe in this process: DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 was not in the "
                                   "environment when HIP initialised (this ROCm overwrites the kernel arguments of instantiated graphs, "
                                   "ultralytics/hip/__init__.py).  Export it, or import ultralytics before the first torch.cuda call; "
                                   "hipgraph=False / use_graph=False runs the same launch lists eagerly.")
        self.use_graph = use_graph
        self.soap = optimizer == "SOAP"  # host-driven (hip/soap.py); the flat kernel then only keeps EMA / loss scale / counters
        self.mode = {"SGD": 0, "Adam": 1, "AdamW": 2, "RMSProp": 3, "RAdam": 4, "Adamax": 5, "NAdam": 6, "SOAP": 0}[optimizer]
        if self.soap and use_graph:
            use_graph = self.use_graph = False  # eigh / QR are not captured; SOAP steps run eagerly
        self.hyper_host = (C.c_float * 16)()  # read by dy_set_hyper at enqueue time (the values travel as kernel arguments)
        if share is not None:
            if share.model is not model or share.mode != self.mode:
                raise ValueError("StepPlan(share=...): the plans must drive the same model with the same optimizer")
            self.crit = share.crit.clone_for_plan()
            self.mom, self.adam_v, self.ema, self.ema_b = share.mom, share.adam_v, share.ema, share.ema_b
            self.hyper, self.state, self.partials = share.hyper, share.state, share.partials
        else:
            self.crit = model.criterion if hasattr(model, "criterion") else model.init_criterion()
            model.criterion = self.crit
            n = self.rt.n_params_flat
            f = lambda: torch.zeros(n, dtype=torch.float32, device=dev)  # noqa: E731
            self.mom = f()
            self.adam_v = f() if self.mode else None
            self.ema = self.rt.flat_p.clone()
            self.ema_b = self.rt.flat_b.clone()
            self.hyper = torch.zeros(16, dtype=torch.float32, device=dev)
            self.state = torch.zeros(8, dtype=torch.float32, device=dev)
            self.state[0] = init_scale
            self.partials = torch.zeros(4096, dtype=torch.float32, device=dev)
        self.input_act = bool(input_act)
        self._fmt_fixed = False  # stage(): the first batch decides the input format
        self.x_in = torch.zeros((batch_size, *self.imgsz, 8), dtype=torch.float16, device=dev) if input_act else None
        self.img = None if input_act else torch.zeros((batch_size, 3, *self.imgsz), dtype=torch.float32, device=dev)
        self.input_u8 = False  # decided by the first batch: uint8 NHWC (the loader's format) or float NCHW (the public tensor API)
        self.flip = None       # (B,) uint8 flip bits when the first loader batch carries them (flips folded into the import kernel)
        self.pool = self.index = None  # HBM-resident image pool + (B,) int32 slots when the loader keeps the dataset on the device
        self.warp = None               # (B,96) int32 mosaic / affine / perspective / MixUp records (two 48-word records per sample): the pool is then read through dy_warp_import_u8
        self.hsv = None                # (B,3) float32 RandomHSV gains applied inside dy_import_image_u8
        self.rec_fb = None
        self.capture_retries = 0  # captures of the step graph that failed their check and were repeated (forward_backward)
        self.rec_opt, self.graph_opt = {}, {}
        self.graph_fb = None
        self.gsum, self._micro = None, 0
        self.ema_updates = 0
        # Data-parallel gradient buckets (DY_DP_BUCKETS=2, world_size > 1): the backward list is cut where the neck + head end; their
        # gradients (bucket 1) are all-reduced on a side stream WHILE the backbone's backward runs, the backbone's (bucket 2) after it
        # -- the reference's DDP reducer fires its buckets inside loss.backward() the same way (engine/trainer.py:695, :810).
        self.buckets = 2 if (world_size > 1 and os.environ.get("DY_DP_BUCKETS", "1") == "2") else 1
        self.fb_cut = None          # index into rec_fb.ops between the two halves of the backward pass
        self._bucket_stream = None
        self._bucket_pending = False
        self._exchange = True
        self.graph_fb2 = None
        self.dynamic_scale = bool(dynamic_scale)  # False (amp=False): the loss scale is a constant, nothing ever halves it
        self.opt_calls = 0                        # optimizer_step() calls so far; the device counts taken + skipped (state[5], state[6])
        self.rt.refresh_frozen()

    # ---- host-side schedule ------------------------------------------------------------------------------------
    def set_hyper(self, lr, momentum, wd, ema_decay=None, max_norm=10.0, beta2=0.999, eps=1e-8):
        h = self.hyper_host
        if ema_decay is None:
            ema_decay = 0.9999 * (1 - math.exp(-(self.ema_updates + 1) / 2000))
        self._hyp = dict(lr=[float(v) for v in lr], momentum=float(momentum), wd=[float(v) for v in wd], max_norm=float(max_norm))
        if self.soap:  # the parameter update happens on the host side; the kernel runs with zero step size
            lr, wd = [0.0] * 3, [0.0] * 3
        vals = [*lr, momentum, *wd, ema_decay, max_norm, beta2, eps, 1.0 if self.dynamic_scale else 0.0]
        for i, v in enumerate(vals):
            h[i] = float(v)
        check(lib().dy_set_hyper(self.hyper.data_ptr(), h, torch.cuda.current_stream(self.hyper.device).cuda_stream), "dy_set_hyper")

    # ---- forward + loss + backward ----------------------------------------------------------------------------
    def _trace_fb(self, batch):
        eng, rt, model, crit = self.eng, self.rt, self.model, self.crit
        eng.rec = Recorder()
        eng.tape = []
        eng.training = True
        eng.arena = self.arena
        if self.arena is not None:
            self.arena.reset()
        try:
            eng.reset_dataflow()
            eng.zero_acc_pool()
            rt.pack_all(transposed=True)
            x = self.import_input()
            from .engine import HEAD_ROWS
            # (the rows / fused-decode / class-kernel forms of the head need their weight gradients on the main stream: with weight
            # gradients on the side stream the trace takes the dense head kernels from the start -- decided HERE, before the forward
            # is traced, so that the loss is never told to leave rows unwritten which a dense backward would then read)
            eng.side_wgrad = self.side_wgrad  # (known while the forward is traced: segmented concatenations need the main-stream weight gradients too)
            head_forms = HEAD_ROWS and not self.side_wgrad
            eng.rows_used, eng.loss_rows = (set() if head_forms else None), None
            eng.pending_decode = [] if head_forms else None
            ho = model.forward_act(x)
            crit.bind(ho, self.nmax, gscale=self.state[0:1])
            crit.__dict__["_last"] = crit
            crit.sync_modes()
            # every level's box conv can be back-propagated from the foreground rows of its gradient: the loss need not zero the rest
            rows = eng.rows_used is not None and len(eng.rows_used) == len(ho.box)
            crit._args.dbox_rows_only = 1 if rows else 0
            crit._args.box_from_input = 0
            if rows:
                eng.loss_rows = crit.assignment_rows()
            if eng.pending_decode:
                # the head left its box logits unwritten: decoded boxes go straight into the loss workspace, the loss recomputes the
                # logits of foreground anchors from the final conv's operands
                assert rows and len(eng.pending_decode) == len(ho.box)
                asg, A, a0 = eng.loss_rows
                a = crit._args
                a.box_from_input = 1
                from .engine import HEAD_BATCH
                P, I = C.c_void_p, C.c_int
                rows_ = []
                for spec, xin, l in eng.pending_decode:
                    xp, xld, xcoef = eng._src(xin)  # the Conv in front may have left its apply out: (raw, ld, coefficient table)
                    rows_.append((xp, xld, xcoef, spec.weight.data_ptr(), spec.bias.data_ptr(), a0[l], xin.H, xin.W))
                    a.box_in[l], a.box_in_ld[l], a.box_in_coef[l] = xp, xld, xcoef
                    a.box_w[l], a.box_b[l] = spec.weight.data_ptr(), spec.bias.data_ptr()
                spec0, x0 = eng.pending_decode[0][0], eng.pending_decode[0][1]
                if HEAD_BATCH:
                    cols = list(zip(*rows_))
                    eng.call("dy_head_box_decode_levels", len(rows_), eng._arr(P, cols[0]), eng._arr(I, cols[1]), eng._arr(P, cols[2]),
                             eng._arr(P, cols[3]), eng._arr(P, cols[4]), crit.pred_box_ptr(), A, eng._arr(I, cols[5]), x0.N,
                             eng._arr(I, cols[6]), eng._arr(I, cols[7]), spec0.cin, spec0.cout)
                else:
                    for xp, xld, xcoef, wp, bp, a0l, h_, w_ in rows_:
                        eng.call("dy_head_box_decode", xp, xld, xcoef, wp, bp, crit.pred_box_ptr(), A, a0l, x0.N, h_, w_, spec0.cin, spec0.cout)
            eng.call("dy_detection_loss", C.byref(crit._args))
            self.fb_split = len(eng.rec.ops)  # [0, fb_split) = forward + loss, the rest = backward (forward_only / backward_accumulate)
            eng.deferred_wgrad = []
            eng.side_wgrad = self.side_wgrad
            mark = eng.tape_mark if (self.buckets == 2 and eng.tape_mark) else 0
            for f in reversed(eng.tape[mark:]):
                f()
            if mark:  # neck + head done: their weight gradients leave the slabs now, the first bucket is complete
                eng.flush_wgrad()
                eng.deferred_wgrad = []
                self.fb_cut = len(eng.rec.ops)
                for f in reversed(eng.tape[:mark]):
                    f()
            eng.flush_wgrad()
        finally:
            eng.deferred_wgrad = None
            eng.rows_used = eng.loss_rows = eng.pending_decode = None
            eng.side_wgrad = False
            eng.acc_zeroed = False
            eng.arena = None
            rec, eng.rec, eng.tape = eng.rec, None, None
        self.ho = ho
        return rec

    def forward_backward(self, batch, exchange=True):
        """Stage the batch, run fwd+loss+bwd; gradients (scaled by the loss scale) land in rt.flat_g.  ``exchange`` (data-parallel runs
        with gradient buckets): the optimizer step follows THIS micro-batch alone -- only then may the first bucket's all-reduce start
        under the backbone's backward, because it sums the gradient buffer IN PLACE (a micro-batch that is going to be accumulated must
        keep its local gradients: pass False)."""
        self._exchange = bool(exchange)
        if not self.input_act:
            self.stage(batch)
        n = self.crit.set_targets(batch, cap=self.B * self.nmax)
        if n > self.B * self.nmax:
            raise RuntimeError(f"{n} targets exceed the plan capacity {self.B}x{self.nmax}")
        self.crit.sync_modes()
        if self.rec_fb is None:
            pre = (self.rt.flat_b.clone(), self.crit.scalars.clone()) if self.use_graph else None
            self.rec_fb = self._trace_fb(batch)
            if self.use_graph:
                rec = self.rec_last = self.rec_fb  # (rec_last: kept for diagnosis when the check below drops rec_fb)
                torch.cuda.synchronize()
                want = (self.crit.scalars.clone(), self.rt.flat_g.clone(), self.rt.flat_b.clone())  # what the TRACED step left
                for attempt in (0, 1):
                    torch.cuda.synchronize()
                    self.rec_fb, self.graph_fb = rec, torch.cuda.CUDAGraph()
                    with torch.cuda.graph(self.graph_fb, capture_error_mode="thread_local"):
                        self.eng.replay(self.rec_fb)
                    graph = self.graph_fb
                    try:
                        self._verify_capture(*pre, want=want)
                        break
                    except RuntimeError as e:
                        # One more independent capture, checked against the same traced reference.  What this absorbs was seen with
                        # two processes sharing one GPU only: ONE execution of the step -- the traced one or a replay, never
                        # the same one twice -- leaves the box-logit gradient of one anchor's bottom side different (box_loss_kernel,
                        # from inputs that are the reference's bits); the next execution is clean again.  The systematic failure the
                        # check exists for (graph packet capture in effect: the FIRST replay is right, one after a burst of launches is
                        # not) repeats, and raises.
                        import warnings
                        if attempt == 0:
                            first = e
                            self.capture_retries += 1
                            warnings.warn(f"StepPlan: capturing the step again -- {e}")
                            continue
                        # Both captures disagree with the traced step on their FIRST replay and agree with EACH OTHER bit for bit:
                        # the traced execution was the odd one out.  The graph is kept; its results are the state.
                        a, b = getattr(first, "got", None), getattr(e, "got", None)
                        if (a is not None and b is not None and "first replay" in getattr(first, "stage", "") and "first replay" in getattr(e, "stage", "")
                                and all(torch.equal(x, y) for x, y in zip(a, b)) and bool(torch.isfinite(b[1]).all())):
                            warnings.warn("StepPlan: two independent captures reproduce each other but not the traced step: keeping the graph")
                            self.rec_fb, self.graph_fb = rec, graph
                            break
                        raise
                if self.fb_cut is not None:  # the same list as two graphs, so that a collective can start between them
                    g1, g2 = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g1, capture_error_mode="thread_local"):
                        self.eng.replay(self.rec_fb, 0, self.fb_cut)
                    with torch.cuda.graph(g2, capture_error_mode="thread_local"):
                        self.eng.replay(self.rec_fb, self.fb_cut, None)
                    self.graph_fb, self.graph_fb2 = g1, g2
            if self.fb_cut is not None:
                self._start_bucket1()  # the traced step ran whole: its first bucket starts now (nothing left to overlap with)
        elif self.fb_cut is not None:
            if self.graph_fb is not None:
                self.graph_fb.replay()
            else:
                self.eng.replay(self.rec_fb, 0, self.fb_cut)
            self._start_bucket1()
            if self.graph_fb2 is not None:
                self.graph_fb2.replay()
            else:
                self.eng.replay(self.rec_fb, self.fb_cut, None)
        elif self.graph_fb is not None:
            self.graph_fb.replay()
        else:
            self.eng.replay(self.rec_fb)
        return self.crit.scalars

    # ---- gradient buckets ------------------------------------------------------------------------------------------------------
    # The flat gradient buffer is laid out [neck + head | backbone | buffer tail] (hip/runtime.py), so a bucket is a VIEW of it: no
    # gather / scatter launches and no staging copies between the two graphs -- RCCL reduces the slices in place.
    def _start_bucket1(self):
        """Neck + head gradients are final: start their all-reduce on the side stream."""
        if self._micro or not self._exchange:  # gradient accumulation exchanges the accumulated buffer once, at the optimizer step
            return
        if self._bucket_stream is None:
            self._bucket_stream = torch.cuda.Stream(self.rt.flat_p.device)
        ready = torch.cuda.Event()
        ready.record()
        with torch.cuda.stream(self._bucket_stream):
            self._bucket_stream.wait_event(ready)
            all_reduce_flat(self.rt.flat_gb[:self.rt.bucket_split], self.world_size)
        self._bucket_pending = True

    # ---- the two halves on their own: ``loss = model(batch); loss.backward()`` (nn/tasks.py, BaseModel.loss) -----------------------
    def forward_only(self, batch):
        """Forward + loss of the staged batch (the first half of the recorded list); returns the loss scalars.  The gradients w.r.t.
        the head outputs are written by the loss kernels at the plan's own loss scale (``state[0]``); nothing else of the backward
        pass runs."""
        self.stage(batch)
        n = self.crit.set_targets(batch, cap=self.B * self.nmax)
        if n > self.B * self.nmax:
            raise RuntimeError(f"{n} targets exceed the plan capacity {self.B}x{self.nmax}")
        self.crit.sync_modes()
        if self.rec_fb is None:  # first call: the trace runs both halves once; the caller's accumulated gradients are put back
            keep = self.rt.flat_g.clone()
            self.rec_fb = self._trace_fb(batch)
            self.rt.flat_g.copy_(keep)
        else:
            self.eng.replay(self.rec_fb, 0, self.fb_split)
        return self.crit.scalars

    def backward_accumulate(self, grad_out):
        """The backward half of the recorded list for the forward that ran last, ADDED into the ``.grad`` views with autograd's
        semantics: ``p.grad += grad_out * dL/dp``.  Internally the pass runs in fp16 at the plan's loss scale; the factor
        grad_out / scale is applied in fp32 while accumulating.  A non-finite result (fp16 overflow) halves the internal scale for
        the next call -- GradScaler's policy -- and is handed on as it is (torch.cuda.amp.GradScaler skips such a step)."""
        g = self.rt.flat_g
        keep = g.clone()
        self.eng.zero_backward_acc()  # the recorded list zeroes the statistic accumulators in its FORWARD half only
        self.eng.replay(self.rec_fb, self.fb_split, None)
        scale = self.state[0:1]
        g.mul_(grad_out.reshape(1).to(g.dtype) / scale).add_(keep)
        if self.dynamic_scale:
            ok = torch.isfinite(g).all()
            scale.copy_(torch.where(ok, scale, scale * 0.5))

    def stage(self, batch):
        """Bring the batch's images into the static buffers the import kernel of this plan reads (``img`` / pool records, flip
        bits, HSV gains); the first call fixes the input format."""
        img = batch["img"]
        first = not self._fmt_fixed
        u8 = img.dtype == torch.uint8 and img.dim() == 4 and img.shape[-1] == 3 and img.shape[1] != 3  # (B,H,W,3)
        if first and u8 and ("index" in batch or "warp" in batch):  # img is the loader's HBM pool: recorded by pointer
            self.pool, self.input_u8 = img, True
            if "warp" in batch:
                self.warp = torch.zeros((self.B, batch["warp"].shape[1]), dtype=torch.int32, device=img.device)
            else:
                self.index = torch.zeros(self.B, dtype=torch.int32, device=img.device)
        if self.pool is not None:
            key = "warp" if self.warp is not None else "index"
            if key not in batch or img.data_ptr() != self.pool.data_ptr():
                raise KeyError(f"this plan was recorded for batches carrying '{key}' records over one HBM-resident image pool")
            (self.warp if self.warp is not None else self.index).copy_(batch[key], non_blocking=True)
            img = self.img  # nothing to stage
        if first and u8 and self.pool is None:
            self.input_u8 = True
            self.img = torch.zeros((self.B, *self.imgsz, 3), dtype=torch.uint8, device=self.img.device)
        if first and u8 and "flip" in batch:
            self.flip = torch.zeros(self.B, dtype=torch.uint8, device=self.img.device)
        if first and u8 and "hsv" in batch and "warp" not in batch:
            self.hsv = torch.ones((self.B, 3), dtype=torch.float32, device=self.img.device)
        if self.hsv is not None:
            self.hsv.copy_(batch["hsv"], non_blocking=True)
        if self.flip is not None:
            if "flip" not in batch:
                raise KeyError("this plan was recorded for batches carrying 'flip' bits (loader with flip_on_device)")
            self.flip.copy_(batch["flip"], non_blocking=True)
        elif "flip" in batch and bool(batch["flip"].any()):
            raise KeyError("the batch carries pending flips but this plan was recorded without them")
        if self.input_u8 and not u8 and self.pool is None:
            raise TypeError("this plan was recorded for uint8 NHWC batches (the loader's format); got "
                            f"{img.dtype} {tuple(img.shape)}")
        if img.data_ptr() != self.img.data_ptr():  # a producer that writes straight into ``plan.img`` (the static input of the
            if u8 and not self.input_u8:           # recorded launch list) skips this staging copy
                img = img.permute(0, 3, 1, 2)
            if img.dtype == torch.uint8 and not self.input_u8:
                img = img.float() / 255
            if tuple(img.shape) != tuple(self.img.shape):
                raise ValueError(f"batch images {tuple(img.shape)} do not match the plan's {tuple(self.img.shape)}")
            self.img.copy_(img, non_blocking=True)
        self._fmt_fixed = True

    def import_input(self):
        """The stem input (fp16 NHWC, 8 channels) of the staged batch: the first launch of the recorded list, or -- called
        eagerly by the multi-scale trainer on the base-size plan -- the image batch to re-interpolate."""
        eng = self.eng
        if self.input_act:
            return eng.wrap_act(self.x_in)
        if self.warp is not None:
            return eng.import_warp(self.pool, self.warp, 8)
        if self.pool is not None:
            return eng.import_image_u8(self.pool, 8, self.flip, self.index, self.hsv)
        if self.input_u8:
            return eng.import_image_u8(self.img, 8, self.flip, None, self.hsv)
        from .engine import ImageAct  # fp32 NCHW (the trainer's ``batch["img"].float() / 255``): a Conv(3->16, 3, 2) stem reads it directly
        return ImageAct(eng, self.img)

    def _verify_capture(self, buffers_before, scalars_before, want=None):
        """Replay the freshly captured forward/backward graph on the traced batch -- from the state the traced step started from
        (BN running statistics, WIoU running mean), so nothing is applied twice -- and require it to reproduce the traced step;
        then once more after a burst of ordinary launches.  A graph can be broken without any error on this ROCm, two ways
        (DESIGN.md section 14): device work issued by ANOTHER host thread while a thread-local capture is open, and -- with the
        runtime's graph packet capture on, which hip/__init__.py turns off at import -- ~1,000 ordinary launches of this library
        between two replays overwriting the arguments of the instantiated graph.  Two eager passes over the recorded list are such
        a burst: if the flag did not take effect (HIP initialised before the import) the second replay fails HERE, not in epoch 2."""
        want_s, want_g, want_b = want if want is not None else (self.crit.scalars.clone(), self.rt.flat_g.clone(), self.rt.flat_b.clone())
        # DY_VERIFY_DUMP=1 (diagnosis): every step-local buffer of the traced step is kept aside; a failing comparison then lists the
        # buffers -- in allocation, i.e. roughly execution, order -- whose contents the replay did not reproduce
        snap = ([(i, t, t.clone()) for i, t in enumerate(self.eng.keep) if torch.is_tensor(t) and t.numel() > 0]
                if os.environ.get("DY_VERIFY_DUMP") == "1" else None)

        def replay_and_compare(what):
            self.rt.flat_b.copy_(buffers_before)
            self.crit.scalars.copy_(scalars_before)
            self.graph_fb.replay()
            torch.cuda.synchronize()
            got_s, got_g = self.crit.scalars, self.rt.flat_g
            ds = float((got_s[5:9] - want_s[5:9]).abs().max() / want_s[5:9].abs().max().clamp_min(1e-12))
            # LDConv's far-sample side pass adds with fp32 atomics: gradients repeat to rounding order only (1e-3); all else is
            # exact.  A traced step that overflowed fp16 (the dynamic loss scale still searching) must overflow again: then only
            # the forward quantities are compared.
            if bool(torch.isfinite(want_g).all()):
                dg = float((got_g - want_g).norm() / want_g.norm().clamp_min(1e-30)) if bool(torch.isfinite(got_g).all()) else float("inf")
            else:
                dg = 0.0 if not bool(torch.isfinite(got_g).all()) else float("inf")
            db = float((self.rt.flat_b - want_b).abs().max() / want_b.abs().max().clamp_min(1e-12))
            # a replay repeats the traced step BIT FOR BIT (fp32 partial sums added in fp64: no order dependence); only LDConv's
            # far-sample side pass (fp32 atomics) leaves rounding-order noise in the gradients
            ld = any(type(mod).__name__ == "LDConv" for mod in self.model.modules())
            # (without LDConv a replay normally repeats the traced step bit for bit; the bound still leaves room for the one case that
            # does not -- an fp64 atomic order flipping the last bit of an fp32 statistic -- and is orders below any real corruption)
            self.capture_exact = ds == 0.0 and dg == 0.0 and db == 0.0
            if not ((ds <= 1e-5 and dg <= 2e-2 and db <= 1e-5) if ld else (ds <= 1e-6 and dg <= 1e-5 and db <= 1e-6)):
                self.graph_fb = self.rec_fb = None
                where = ""
                if snap is not None:
                    print(f"[verify dump] scalars traced {[f'{v:.9g}' for v in want_s.tolist()]}\n[verify dump] scalars replay {[f'{v:.9g}' for v in got_s.tolist()]}"
                          f"\n[verify dump] state {[f'{v:.9g}' for v in self.state.tolist()]}", file=sys.stderr, flush=True)
                    bad = []
                    for i, t, c in snap:
                        a_, b_ = t.view(-1).view(torch.uint8), c.view(-1).view(torch.uint8)
                        if not torch.equal(a_, b_):
                            tf, cf = t.float().view(-1), c.float().view(-1)
                            nz = (tf != cf)
                            idx = int(nz.nonzero()[0]) if bool(nz.any()) else -1
                            bad.append(f"#{i} {tuple(t.shape)} {str(t.dtype)[6:]} differing {int(nz.sum())} first@{idx} max|d| {float((tf - cf).abs().nan_to_num(1e30).max()):.3e}")
                            if int(nz.sum()) <= 4096 and t.dtype != torch.uint8:
                                ii = nz.nonzero().view(-1)[:24].tolist()
                                bad.append("      " + " ".join(f"[{k}] {float(cf[k]):.4g}->{float(tf[k]):.4g}" for k in ii))
                    print(f"[verify dump] {len(bad)} of {len(snap)} kept buffers differ:\n  " + "\n  ".join(bad[:40]), file=sys.stderr, flush=True)
                    refp = os.environ.get("DY_VERIFY_REF")  # the same buffers saved by a clean process (scratch tooling): is it the TRACED
                    if refp and os.path.exists(refp):       # step or the replay that left the reference?
                        ref = torch.load(refp)
                        for tag, pick in (("traced", lambda t, c: c), ("replay", lambda t, c: t)):
                            out = []
                            for i, t, c in snap:
                                r = ref.get(f"keep{i}")
                                v = pick(t, c).detach().cpu()
                                if r is None or r.shape != v.shape or r.dtype != v.dtype or v.dtype == torch.uint8 or v.dtype == torch.float64:
                                    continue
                                if not torch.equal(v.view(-1).view(torch.uint8), r.view(-1).view(torch.uint8)):
                                    vf, rf = v.float().view(-1), r.float().view(-1)
                                    nz = (vf != rf).nonzero().view(-1)
                                    out.append(f"#{i} {tuple(v.shape)} differing {nz.numel()} first@{int(nz[0]) if nz.numel() else -1} "
                                               + " ".join(f"[{k}] {float(rf[k]):.7g}->{float(vf[k]):.7g}" for k in nz[:6].tolist()))
                            print(f"[verify dump] {tag} step vs the clean reference: {len(out)} buffers differ\n  " + "\n  ".join(out[:12]), file=sys.stderr, flush=True)
                        # the loss's workspace (pred_box, targets, assignment, scores ...) and the logit gradients as the failing replay left them
                        extra = [("loss_ws", self.crit._ws)] + [(f"dbox{l}", t) for l, t in enumerate(getattr(self.ho, "dbox", None) or [])]
                        for n, t in extra:
                            r = ref.get(n)
                            if r is None or t is None:
                                continue
                            a_, b_ = t.detach().cpu().contiguous().view(-1).view(torch.uint8), r.contiguous().view(-1).view(torch.uint8)
                            nzb = (a_ != b_).nonzero().view(-1) if a_.numel() == b_.numel() else None
                            print(f"[verify dump] {n} after the failing replay vs the clean reference: "
                                  + ("size mismatch" if nzb is None else f"{nzb.numel()} bytes differ"
                                     + (f" first@{int(nzb[0])} last@{int(nzb[-1])}" if nzb.numel() else "")), file=sys.stderr, flush=True)
                if dg > 0 and bool(torch.isfinite(got_g).all()):  # which parameters' gradients differ (the three largest)
                    offs = sorted((o, n) for n, o in self.rt.param_off.items())
                    d = (got_g - want_g).abs()
                    worst = []
                    for k, (o, n) in enumerate(offs):
                        e = offs[k + 1][0] if k + 1 < len(offs) else d.numel()
                        m = float(d[o:e].max()) if e > o else 0.0
                        if m > 0:
                            worst.append((m / max(float(want_g[o:e].abs().max()), 1e-30), n))
                    where = "; gradients that differ: " + ", ".join(f"{n} ({m:.1e})" for m, n in sorted(worst, reverse=True)[:(len(worst) if os.environ.get("DY_VERIFY_VERBOSE") else 3)]) + f" of {len(worst)}"
                err = RuntimeError(f"the captured step graph does not reproduce the traced step {what} (loss items off by {ds:.2e}, "
                                   f"gradients by {dg:.2e}, BN statistics by {db:.2e} relative{where}): was another host thread issuing device "
                                   "work during the capture, or was HIP initialised before `import ultralytics` could set "
                                   "DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 (export it, or import the package before the first torch.cuda call)?")
                err.stage, err.got = what, (got_s.clone(), got_g.clone(), self.rt.flat_b.clone())  # (forward_backward's second opinion)
                raise err

        replay_and_compare("on its first replay")
        if os.environ.get("DY_VERIFY_BURST") != "0":  # "0": tools/graph_packet_capture.py, to show the corruption happening later
            for _ in range(2):
                self.eng.replay(self.rec_fb)
            replay_and_compare("after 2 eager passes over the same launch list")

    # ---- optimizer ----------------------------------------------------------------------------------------------
    def all_reduce(self):
        """The step's ONE collective: RCCL all-reduce(SUM) over [flat fp32 gradients | float buffers] (4 MB + 25 KB for
        DEAL-YOLO-N).  The buffer tail makes rank 0's BatchNorm running statistics win on every rank, as DDP's
        broadcast_buffers does before each forward (reference engine/trainer.py:640-651): rank 0 stages its buffers, every other
        rank zeros, so the sum IS rank 0's copy.  (Training-mode forwards never read running statistics, so when the exchange
        happens within the step is unobservable; rank 0's own statistics are never overwritten by another rank's.)"""
        if self.world_size > 1:
            rt, n = self.rt, self.rt.n_params_flat
            t = self.gsum if self._micro else rt.flat_gb
            if dist.get_rank() == 0:
                t[n:].copy_(rt.flat_b)
            else:
                t[n:].zero_()
            if self._bucket_pending and not self._micro:
                # bucket 1 is (being) reduced on the side stream; bucket 2 = the backbone's gradients + the buffer tail goes now, in place
                all_reduce_flat(t[rt.bucket_split:], self.world_size)
                torch.cuda.current_stream().wait_stream(self._bucket_stream)
                self._bucket_pending = False
            else:
                if self._bucket_pending:  # an accumulating step: the early bucket is not used
                    torch.cuda.current_stream().wait_stream(self._bucket_stream)
                    self._bucket_pending = False
                all_reduce_flat(t, self.world_size)
            rt.flat_b.copy_(t[n:])

    def accumulate(self):
        """Gradient accumulation across micro-batches (reference engine/trainer.py:812: step only every ``accumulate``
        iterations): fold this micro-step's gradients into the running sum the optimizer will read."""
        if self.gsum is None:
            self.gsum = torch.zeros_like(self.rt.flat_gb)  # same [gradients | buffer staging] layout as the exchange buffer
        n = self.rt.n_params_flat
        if self._micro == 0:
            self.gsum[:n].copy_(self.rt.flat_g)
        else:
            self.eng.call("dy_axpy_f32", self.gsum.data_ptr(), self.rt.flat_g.data_ptr(), 1.0, n)
        self._micro += 1

    def _soap_step(self, grads):
        """optimizer='SOAP' (reference engine/trainer.py:1156-1165): unscale + global-norm clip like ``optimizer_step`` does, then
        hip/soap.py over views of the flat parameter buffer.  Synchronises (non-finite gradients skip the step, as GradScaler does)."""
        rt = self.rt
        self._soap_make()
        g = grads[:rt.n_params_flat]
        if not bool(torch.isfinite(g).all()):
            return  # the kernel below sees the same non-finite gradients and counts the skip
        scale = float(self.state[0])
        norm = float(g.norm()) / scale
        coef = min(self._hyp["max_norm"] / (norm + 1e-6), 1.0) / scale
        self._soap.step([g[o:o + k].view(sh) * coef for _, o, k, sh, _ in self._soap_views], self._hyp["lr"], self._hyp["wd"])

    def _soap_make(self):
        rt = self.rt
        if getattr(self, "_soap", None) is None:
            from .soap import Soap
            views = []
            for n, p in self.model.named_parameters():
                if p.requires_grad:
                    views.append((n, rt.param_off[n], p.numel(), tuple(p.shape), rt.param_group[n]))
            self._soap_views = views
            self._soap = Soap([(rt.flat_p[o:o + k].view(sh), g) for _, o, k, sh, g in views], beta1=self._hyp["momentum"], beta2=0.95)
            self._soap_apply_pending()
        return self._soap

    def soap_state(self):
        """SOAP's per-parameter state (step, moments, Gram matrices, eigenbases) keyed by parameter name, on the host (resume)."""
        so = getattr(self, "_soap", None)
        if so is None:
            return {}
        out = {}
        for i, (n, *_r) in enumerate(self._soap_views):
            st = so.state.get(i)
            if st is not None:
                cpu = lambda t: None if t is None else t.detach().cpu().clone()  # noqa: E731
                out[n] = dict(step=st.step, m=cpu(st.m), v=cpu(st.v), gg=[cpu(t) for t in st.gg], q=None if st.q is None else [cpu(t) for t in st.q])
        return out

    def load_soap_state(self, saved):
        """Resume: applied when the optimizer object is built (its betas come from the first ``set_hyper``)."""
        self._soap_pending = dict(saved)
        if getattr(self, "_soap", None) is not None:
            self._soap_apply_pending()

    def _soap_apply_pending(self):
        saved, self._soap_pending = getattr(self, "_soap_pending", None), None
        if not saved:
            return
        from .soap import _State
        dev = self.rt.flat_p.device
        mv = lambda t: None if t is None else t.to(dev)  # noqa: E731
        for i, (n, o, k, sh, _g) in enumerate(self._soap_views):
            if n in saved:
                d = saved[n]
                st = self._soap.state.get(i) or _State(self.rt.flat_p[o:o + k].view(sh))
                st.step, st.m, st.v, st.gg = d["step"], mv(d["m"]), mv(d["v"]), [mv(t) for t in d["gg"]]
                st.q = None if d["q"] is None else [mv(t) for t in d["q"]]
                self._soap.state[i] = st

    def optimizer_step(self):
        rt, eng = self.rt, self.eng
        grads = self.gsum if self._micro else rt.flat_g
        if self.soap:
            self._soap_step(grads)
        key = grads.data_ptr()
        if key not in self.rec_opt:
            eng.rec = Recorder()
            try:
                b = (C.c_long * 5)(*rt.seg_bounds)  # (read at enqueue time: travels as kernel arguments)
                eng.call("dy_optimizer_step_seg", rt.flat_p.data_ptr(), grads.data_ptr(), self.mom.data_ptr(),
                         0 if self.adam_v is None else self.adam_v.data_ptr(), self.ema.data_ptr(), rt.n_params_flat, b,
                         rt.frozen.data_ptr(), rt.flat_b.data_ptr(), self.ema_b.data_ptr(), rt.n_buffers_flat,
                         self.hyper.data_ptr(), self.state.data_ptr(), self.partials.data_ptr(), self.mode)
            finally:
                self.rec_opt[key], eng.rec = eng.rec, None
            if self.use_graph:
                torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, capture_error_mode="thread_local"):
                    eng.replay(self.rec_opt[key])
                self.graph_opt[key] = g
        elif key in self.graph_opt:
            self.graph_opt[key].replay()
        else:
            eng.replay(self.rec_opt[key])
        self.ema_updates += 1
        self.opt_calls += 1
        self._micro = 0
        rt.mark_dirty()

    def check_progress(self):
        """Fail loudly when optimizer steps do not take effect (synchronises; the trainer calls it where it already does: at
        epoch end).  The device counts the steps it applied (state[5]) and the ones it skipped for non-finite gradients
        (state[6]); a run whose steps are all skipped trains nothing and would otherwise only show as a flat loss curve."""
        st = self.state.cpu().tolist()
        scale, taken, skipped = st[0], int(st[5]), int(st[6])
        if taken + skipped != self.opt_calls:
            raise RuntimeError(f"optimizer step counter out of step with the host: {self.opt_calls} optimizer_step() calls, device took "
                               f"{taken} and skipped {skipped} -- the recorded optimizer launches are not executing")
        if not self.dynamic_scale and skipped:
            raise RuntimeError(f"{skipped} of {self.opt_calls} optimizer steps were skipped for non-finite gradients at the fixed loss "
                               f"scale {scale:g} (amp=False): the gradients are corrupt or overflow fp16 -- nothing was learned from them")
        if self.dynamic_scale and (scale < 1.0 or (self.opt_calls >= 32 and taken == 0)):
            raise RuntimeError(f"dynamic loss scale collapsed to {scale:g} ({skipped} of {self.opt_calls} optimizer steps skipped): "
                               "gradients stay non-finite however small the scale")
        self.crit.check_capacity()
        return taken, skipped, scale

    def step(self, batch, lr, momentum, wd):
        """forward/backward + gradient all-reduce + optimizer/EMA: one iteration at accumulate == 1."""
        self.set_hyper(lr, momentum, wd)
        self.forward_backward(batch)
        self.all_reduce()
        self.optimizer_step()

    def loss_items(self):
        """(loss*B, [box, cls, dfl]) -- synchronises."""
        s = self.crit.scalars.cpu()
        self.crit.check_capacity(s)
        return float(s[8]), s[5:8].clone()

    # ---- measurement helpers (bench.py) -------------------------------------------------------------------------
    def profile_ops(self, reps=3):
        """Event-timed, un-captured replay of the forward/backward launch list on the launch stream.
        Returns [(name, args, avg_ms)] per recorded C-ABI call."""
        ops = [o for o in self.rec_fb.ops if o[0] is not None]  # everything on one stream here: fork/join markers dropped
        s = self.eng.stream
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(len(ops) + 1)]
        tot = [0.0] * len(ops)
        for _ in range(reps):
            # park the GPU for a few tens of ms so that the whole launch list is queued before the first kernel starts: the
            # event intervals then hold kernel time + the back-to-back dispatch gap, not the host's ctypes launch latency
            torch.cuda._sleep(60_000_000)
            evs[0].record()
            for i, (fn, args, name, _side) in enumerate(ops):
                rc = fn(*args, s)
                if rc != 0:
                    check(rc, name)
                evs[i + 1].record()
            torch.cuda.synchronize()
            for i in range(len(ops)):
                tot[i] += evs[i].elapsed_time(evs[i + 1])
        # an event record is itself a queue packet: the interval between two records with nothing between them is not zero.
        # Measure that empty interval in the same queued state and take it off every launch's interval.
        cal = [torch.cuda.Event(enable_timing=True) for _ in range(33)]
        torch.cuda._sleep(20_000_000)
        for e in cal:
            e.record()
        torch.cuda.synchronize()
        gaps = sorted(cal[i].elapsed_time(cal[i + 1]) for i in range(32))
        self.event_gap_ms = gaps[len(gaps) // 2]
        return [(ops[i][2], ops[i][1], max(tot[i] / reps - self.event_gap_ms, 0.0)) for i in range(len(ops))]

    @staticmethod
    def conv_algorithmic_bytes(args):
        """Algorithmic HBM bytes of one dy_conv_forward launch: input read once + output written once (+ read when
        accumulating), at the storage dtype (SURVEY.md 8(d) definition)."""
        (_x, _ldx, _w, _b, _y, _ldy, _p, n, h, w, cin, cout, ks, stride, dil, oh, ow, epi, _np) = args
        H, W = (2 * h, 2 * w) if dil == 2 else (h, w)
        pad = ks // 2
        Ho, Wo = ((H + 2 * pad - ks) // stride + 1, (W + 2 * pad - ks) // stride + 1) if not oh else (oh, ow)
        ob = 4 if epi & 8 else 2
        out = n * Ho * Wo * cout * ob
        return n * h * w * cin * 2 + out * (2 if epi & 16 else 1)

    @staticmethod
    def wgrad_algorithmic_bytes(args):
        """Algorithmic HBM bytes of one dy_conv_wgrad launch: the layer input and the output gradient read once each at the
        storage dtype (the fp32 weight gradient is negligible) -- SURVEY.md 8(d): training = 3 x forward."""
        (_x, _ldx, _dy, _lddy, _sl, _dw, n, h, w, cin, cout, ks, stride, _acc) = args[:14]
        pad = ks // 2
        Ho, Wo = (h + 2 * pad - ks) // stride + 1, (w + 2 * pad - ks) // stride + 1
        return n * h * w * ((cin + 7) // 8 * 8) * 2 + n * Ho * Wo * ((cout + 7) // 8 * 8) * 2

    def kernel_of(self, name, args):
        """(kernel name as rocprofv3 prints it, algorithmic bytes, the launch's own least traffic) of one recorded C-ABI call.
        Under SURVEY.md 8(d)'s definition only the convolutions (forward, input gradient, weight gradient) have algorithmic
        bytes: BatchNorm / activation passes and every other element-wise launch count as fused away, i.e. 0 -- that is what the
        STEP's roofline is priced with.  A kernel's OWN roofline needs the bytes that launch cannot avoid (each operand read once,
        each result written once at the storage dtype): equal to the algorithmic bytes for a convolution, the streams of the pass
        for a BatchNorm / activation kernel (which 8(d) prices at 0 because a perfect fusion would not launch it at all)."""
        key, alg = self._kernel_alg(name, args)
        own = alg
        e = 2  # fp16 storage
        if name == "dy_bn_act_apply":
            own = int(args[7]) * int(args[8]) * e * (3 if args[2] else 2)
        elif name == "dy_bn_act_apply_acc":
            own = int(args[12]) * int(args[13]) * e * (3 if args[2] else 2)
        elif name == "dy_bn_act_bwd_reduce":
            own = int(args[7]) * int(args[8]) * e * 2
        elif name == "dy_bn_act_bwd_reduce_acc":  # dy and raw read; the shortcut gradient stored (or read, added and stored)
            own = int(args[6]) * int(args[7]) * e * (2 + ((2 if args[11] else 1) if args[9] else 0))
        elif name == "dy_bn_act_bwd_apply":
            own = int(args[8]) * int(args[9]) * e * 3
        elif name == "dy_bn_act_bwd_apply_acc":
            own = int(args[10]) * int(args[11]) * e * 3
        elif name in ("dy_conv_wgrad_bn", "dy_conv_wgrad_ld_bn"):  # + raw read and d(raw) written, both the size of the output gradient
            n, h, w = args[14:17]
            if name == "dy_conv_wgrad_bn":
                cin, cout, ks, stride = args[17:21]
            else:
                cout, ks, stride = args[17], 1, 1
            pad = ks // 2
            own = alg + 2 * n * ((h + 2 * pad - ks) // stride + 1) * ((w + 2 * pad - ks) // stride + 1) * ((cout + 7) // 8 * 8) * e
        elif name == "dy_stem_wgrad_bn":
            n, h, w = args[11:14]
            own = alg + n * ((h - 1) // 2 + 1) * ((w - 1) // 2 + 1) * 16 * e
        elif name in ("dy_conv1x1_wgrad_bn_segs", "dy_conv1x1_wgrad_bn_planes") and alg:  # + raw read and d(raw) written
            n, h, w, _cin, cout = args[13:18] if name.endswith("segs") else args[17:22]
            own = alg + 2 * n * h * w * ((cout + 7) // 8 * 8) * e
        elif name == "dy_bn_act_apply_acc_split":
            own = int(args[13]) * int(args[14]) * e * 2
        elif name == "dy_bn_act_bwd_reduce_acc_split":
            own = int(args[9]) * int(args[10]) * e * 2
        return key, alg, own

    def _kernel_alg(self, name, args):
        import ctypes as C
        L, buf = self.eng.L, C.create_string_buffer(128)
        if name == "dy_conv_forward":
            n, h, w, cin, cout, ks, stride, dil = args[7:15]
            if dil == 2 and ks == 3:
                return f"conv_mfma_dg2_kernel (input gradient of the {cout}->{cin} stride-2 3x3)", self.conv_algorithmic_bytes(args)
            pad = ks // 2
            if L.dy_conv_kernel_name_at(cin, cout, ks, stride, (w + 2 * pad - ks) // stride + 1, dil, int(args[17]), buf, 128) == 0:
                return buf.value.decode(), self.conv_algorithmic_bytes(args)
        # 1x1 convolutions over a never-materialised concatenation (the segment table travels by reference: args[k]._obj)
        if name == "dy_conv1x1_forward_segs":
            n, h, w, cin, cout = args[6:11]
            if L.dy_conv1x1_segs_kernel_name(cin, cout, args[0], buf, 128) == 0:
                return buf.value.decode(), n * h * w * (cin + (cout + 7) // 8 * 8) * 2
        if name == "dy_conv1x1_input_grad_segs":  # (dy, lddy, w^T, dxs, n, h, w, channels of dy, channels of the concatenation)
            n, h, w, cin, cout = args[4:9]
            if L.dy_conv_kernel_name(cin, cout, 1, 1, buf, 128) == 0:
                return buf.value.decode(), n * h * w * (cin + cout) * 2
        if name in ("dy_conv1x1_wgrad_bn_segs", "dy_conv1x1_wgrad_bn_planes"):
            n, h, w, cin, cout = args[13:18] if name.endswith("segs") else args[17:22]
            if L.dy_wgrad_kernel_name_at(n, h, w, cin, cout, 1, 1, buf, 128) == 0:
                seg = args[0] is not None
                return buf.value.decode().replace(", 0>", ", 3>" if seg else ", 1>"), n * h * w * (cin + (cout + 7) // 8 * 8) * 2
        if name == "dy_conv_wgrad":
            n, h, w, cin, cout, ks, stride = args[6:13]
            if L.dy_wgrad_kernel_name_at(n, h, w, cin, cout, ks, stride, buf, 128) == 0:
                return buf.value.decode(), self.wgrad_algorithmic_bytes(args)
        if name == "dy_conv_input_grad_red":  # (dy, lddy, w, dx, lddx, n, h, w, cin, cout, ks, ...): a stride-1 input gradient
            n, h, w, cin, cout, ks = args[5:11]
            if L.dy_conv_kernel_name(cin, cout, ks, 1, buf, 128) == 0:
                return buf.value.decode().replace("false, 0>", "true, 0>"), n * h * w * (cin + cout) * 2
        if name == "dy_stem_forward":  # (img, w, raw, ldraw, acc, n, h, w, mul): 3 input channels read once, 16 output channels written once
            n, h, w = args[5:8]
            return "stem_fwd_kernel", n * h * w * 3 * 2 + n * ((h - 1) // 2 + 1) * ((w - 1) // 2 + 1) * 16 * 2
        if name == "dy_stem_wgrad_bn":
            n, h, w = args[11:14]
            return "stem_wgrad_bn_kernel", n * h * w * 3 * 2 + n * ((h - 1) // 2 + 1) * ((w - 1) // 2 + 1) * 16 * 2
        if name in ("dy_conv_wgrad_bn", "dy_conv_wgrad_ld_bn"):  # BatchNorm backward apply inside the kernel: d(raw) is written too
            n, h, w = args[14:17]
            if name == "dy_conv_wgrad_bn":
                cin, cout, ks, stride = args[17:21]
            else:
                cout, _ld_cin, ld_taps, ld_cphys = args[17:21]
                cin, ks, stride = ld_taps * ld_cphys, 1, 1
            if L.dy_wgrad_kernel_name_at(n, h, w, cin, cout, ks, stride, buf, 128) == 0:
                pad = ks // 2
                Ho, Wo = (h + 2 * pad - ks) // stride + 1, (w + 2 * pad - ks) // stride + 1
                return (buf.value.decode().replace(", 0>", ", 1>"),
                        n * h * w * ((cin + 7) // 8 * 8) * 2 + n * Ho * Wo * ((cout + 7) // 8 * 8) * 2)
        if name == "dy_conv_wgrad_ld":
            n, h, w, cout, ld_cin, ld_taps, ld_cphys = args[6:13]
            if L.dy_wgrad_kernel_name_at(n, h, w, ld_taps * ld_cphys, cout, 1, 1, buf, 128) == 0:
                return buf.value.decode(), n * h * w * (ld_taps * ld_cphys + (cout + 7) // 8 * 8) * 2
        tmpl = {"dy_bn_act_apply": ("bn_act_apply_kernel<{}, false, false>", 9), "dy_bn_act_apply_acc": ("bn_act_apply_kernel<{}, true, " + ("false" if os.environ.get("DY_SILU_FAST") == "0" else "true") + ">", 14),
                "dy_bn_act_bwd_reduce": ("bn_act_bwd_reduce_kernel<{}, false>", 9),
                "dy_bn_act_bwd_apply": ("bn_act_bwd_apply_kernel<{}, false>", 10),
                "dy_bn_act_bwd_apply_acc": ("bn_act_bwd_apply_kernel<{}, true>", 12)}
        if name == "dy_bn_act_bwd_reduce_acc":  # <activation, shortcut gradient passed on>
            return f"bn_act_bwd_reduce_kernel<{int(args[8])}, {'true' if args[9] else 'false'}>", 0
        if name == "dy_bn_act_bwd_reduce_acc_split":
            return f"bn_act_bwd_reduce_kernel<{int(args[11])}, false>", 0
        if name == "dy_bn_act_apply_acc_split":
            return f"bn_act_apply_kernel<{int(args[15])}, true, true>", 0
        if name in tmpl:
            k, i = tmpl[name]
            return k.format(int(args[i])), 0
        return name.replace("dy_", "", 1) + " (C-ABI call)", 0

    def probe_dominant_kernel(self, batch, reps=10):
        """Time every launch of the step (HIP events on the launch stream), group the launches by the kernel they run -- ALL
        launches, whatever they are -- and price the group with the largest total time per step: average launch duration and
        average ALGORITHMIC bytes per launch over all its launches of one step (what `rocprofv3 --kernel-trace --stats` averages
        for the same kernel name).  The top conv instantiation is reported beside it."""
        if self.rec_fb is None:
            self.forward_backward(batch)
        prof = self.profile_ops(reps)
        self.last_profile = prof
        groups = {}
        for name, args, ms in prof:
            key, by, own = self.kernel_of(name, args)
            groups.setdefault(key, []).append((ms, by, own))

        def summary(key):
            items = groups[key]
            ms = sum(t[0] for t in items) / len(items)
            by = sum(t[2] for t in items) / len(items)  # the kernel's own least traffic (== algorithmic bytes for a convolution)
            return {"kernel": key, "us": ms * 1e3, "bytes": by, "gbs": by / (ms * 1e-3) / 1e9 if ms > 0 else 0.0,
                    "algorithmic_bytes": sum(t[1] for t in items) / len(items),
                    "launches_per_step": len(items), "ms_per_step": sum(t[0] for t in items)}

        if not groups:
            return None
        rank = sorted(groups, key=lambda k: -sum(t[0] for t in groups[k]))
        top = summary(rank[0])
        convs = [k for k in rank if k.startswith("conv_")]
        top["top_conv"] = summary(convs[0]) if convs else None
        # the dominant kernel in SURVEY.md 8(d)'s sense: the group with the largest total time among the launches that HAVE algorithmic
        # bytes (convolutions: forward, input gradient, weight gradient, stem); priced with those bytes alone
        algs = [k for k in rank if sum(t[1] for t in groups[k]) > 0]
        if algs:
            ta = summary(algs[0])
            ta["alg_gbs"] = ta["algorithmic_bytes"] / (ta["us"] * 1e-6) / 1e9 if ta["us"] > 0 else 0.0
            top["top_alg"] = ta
        else:
            top["top_alg"] = None
        top["ranking"] = [{"kernel": k, "ms_per_step": round(sum(t[0] for t in groups[k]), 4), "launches": len(groups[k]),
                           "algorithmic_MB_per_step": round(sum(t[1] for t in groups[k]) / 1e6, 1),
                           "own_MB_per_step": round(sum(t[2] for t in groups[k]) / 1e6, 1)} for k in rank[:12]]
        top["step_device_ms"] = sum(t[2] for t in prof)
        top["step_algorithmic_bytes"] = sum(t[1] for v in groups.values() for t in v)
        return top

    def breakdown(self):
        agg = {}
        for name, _a, ms in getattr(self, "last_profile", []):
            k = agg.setdefault(name, [0, 0.0])
            k[0] += 1
            k[1] += ms
        return {k: {"calls": v[0], "ms": round(v[1], 3)} for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])}
