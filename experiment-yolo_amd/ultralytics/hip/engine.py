"""Execution engine of the MI355X DEAL-YOLO path: NHWC fp16 activations, a tape of hand-written backward launches,
and record/replay of the whole step as a static launch list (capturable into one hipGraph).

Design:
  * ``Storage`` = one contiguous (N,H,W,C) fp16 device buffer plus a lazily created gradient twin; ``Act`` = a channel
    slice [c0, c0+C) of a Storage.  Concat/chunk are therefore free: producers write into slices, consumers read them.
  * every kernel launch goes through ``Engine.call``; while a ``Recorder`` is active the (function, ctypes args) pair
    is also appended to a list, so the first (traced) step leaves behind the exact launch sequence of forward, loss,
    backward and optimizer, which later steps replay without touching Python module code or the allocator.
  * backward is NOT torch autograd: each forward op pushes a closure that issues the backward kernels; gradient fan-in
    is resolved statically (first writer stores, later writers accumulate).
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np
import torch

from . import (DY_ACT_LEAKY, DY_ACT_NONE, DY_ACT_SILU, DY_BN_COPIES, DY_EPI_ACCUM, DY_EPI_BIAS, DY_EPI_F32OUT, DY_EPI_SILU,
               DY_EPI_STATS, DY_EPI_STATS_ACC, DySegs, check, lib)

BN2D_EPS, BN2D_MOM = 1e-3, 0.03  # reference utils/torch_utils.py:347-349
BN3D_EPS, BN3D_MOM = 1e-5, 0.1  # nn.BatchNorm3d defaults (ScalSeq), untouched by initialize_weights


def _ptr(t):
    return 0 if t is None else t.data_ptr()


# Debug switch (tests/test_gpu_poison.py, env DY_POISON=1): every buffer the engine hands out uninitialised is filled with
# 0xFF bytes -- NaN as fp16 / fp32, -1 as an integer -- so that a kernel reading memory nothing wrote shows up as a NaN in
# the losses / gradients instead of depending on what the caching allocator happened to return.
POISON = os.environ.get("DY_POISON", "0") == "1"

# BatchNorm statistics without finalize launches (round 3): the conv epilogue / the backward reduce add their per-workgroup sums
# into fp64 accumulators and the apply kernels sum them in their prologue (csrc/bn_act.hip, BnAccFwd).  DY_BN_ACC=0 restores the
# three-launch form (partial rows -> dy_bn_finalize -> apply), kept for measurement and for ScalSeq's three-resolution statistics.
BN_ACC = os.environ.get("DY_BN_ACC", "1") != "0"
# ... and the backward apply pass folded into the weight-gradient kernel (csrc/conv_wgrad.hip, BNF): the kernel forms d(raw) from
# (dy, raw) while it stages its dY operand and writes it once for the input-gradient pass.  DY_BN_WGRAD=0: separate apply launch.
BN_WGRAD = BN_ACC and os.environ.get("DY_BN_WGRAD", "1") != "0"
# The stem Conv(3 -> 16, k 3, s 2) reads the fp32 NCHW image batch directly (csrc/stem.hip) instead of an imported fp16 copy padded to
# 8 channels.  DY_STEM_DIRECT=0: import kernel + generic conv / weight-gradient kernels.
STEM_DIRECT = BN_WGRAD and os.environ.get("DY_STEM_DIRECT", "1") != "0"
# The gradient of a Bottleneck shortcut (= dy) stored / added by the backward reduce pass while dy streams through it, instead of by
# a dy_add / dy_copy_slice launch.  DY_BN_RES=0: separate launch.
BN_RES = BN_ACC and os.environ.get("DY_BN_RES", "1") != "0"
# Bias gradients (Detect's final convs, LDConv.p_conv) summed inside the weight-gradient kernel and finished by the batched slab
# reduction, instead of a reduce + finalize launch pair per layer.  DY_BIAS_WGRAD=0: the launch pair.
BIAS_WGRAD = BN_ACC and os.environ.get("DY_BIAS_WGRAD", "1") != "0"
# The first pass of a Conv's BatchNorm backward (sums of g and g*xhat) run by the input-gradient kernel that is the ONLY writer of that
# Conv's output gradient, in its epilogue (csrc/conv.hip, REDK), instead of by a dy_bn_act_bwd_reduce_acc launch that re-reads the
# gradient.  MEASURED SLOWER (round 3, one box, alternating): 12.87 ms per step without, 13.06 ms with it for outputs up to 32 channels,
# 15.34 ms up to 64 channels (those instantiations spill): exp + rcp per element in the memory slot of the ping-pong kernel make that
# slot, not the MFMA slot, set the pace (a fused 32->32 3x3 dgrad at 160x160: 90-107 us against ~50 us + ~50 us for the two launches).
# Off by default; DY_BN_DGRED=1 enables it, DY_BN_DGRED_MAXC bounds the output width it is used for, DY_BN_DGRED_MAXPIX the map size
# (round 4: restricted to the latency-bound maps <= 80x80 it still loses, 11.44 -> 11.48-11.51 ms: profiles/r04_stage_ab.md).
# SPPF's three chained 5x5 pools (and their backward chain) as one launch each with the map resident in LDS, when it fits
# (Engine.sppf_pools); maps that do not fit take three dy_maxpool5 / dy_maxpool5_backward launches (the form a test flips this to).
SPPF_FUSED = True
# C2f.cv1's two halves (reference block.py:223, ``cv1(x).chunk(2, 1)``) as tensors of their own -- what planar concatenations left
# sliced: the second half is what the first Bottleneck reads, adds (shortcut) and back-propagates into.  The apply pass writes both
# planes, the backward reduce and the weight-gradient kernel read the gradient from both (dy_*_split / dy_conv1x1_wgrad_bn_planes).
# DY_PLANAR_CV1=0: cv1 writes one 2c-wide tensor, the second half is a channel slice of it.
PLANAR_CV1 = os.environ.get("DY_PLANAR_CV1", "1") != "0"
# Inference: nn.Upsample(None, 2, 'nearest') in front of a Concat (every model YAML's top-down path) is never executed -- the 1x1 conv
# behind the concatenation reads pixel (y >> 1, x >> 1) of the low-resolution tensor where it stages that member (UpAct; DySegs.acc = 2).
# DY_UPSEG=0: dy_upsample2x writes the four-times-larger copy.
UPSEG = os.environ.get("DY_UPSEG", "1") != "0"
# Weight gradients of SMALL maps (N*H*W <= DY_SIDE_SMALL pixels) on the side stream beside the input-gradient chain: there a launch is
# mostly fixed latency and leaves CUs idle (profiles/r04_stage_ab.md), so the layer gives up the BatchNorm-in-the-weight-gradient fusion
# (a separate backward-apply launch writes d(raw)) to take its weight gradient off the critical path.  0 (default): off -- MEASURED
# SLOWER (round 4, one box): 11.33 ms off, 11.66 for maps <= 40x40, 11.99 <= 80x80, 12.17 <= 160x160 (profiles/r04_stage_ab.md).
SIDE_SMALL = int(os.environ.get("DY_SIDE_SMALL", "0"))
# ... in training too: forward and weight gradient read the low-resolution tensor, the input gradient of that member goes to a
# full-resolution gradient tensor that Upsample's backward folds as before.  DY_UPSEG_TRAIN=0: training runs the up-sampling launch.
UPSEG_TRAIN = UPSEG and os.environ.get("DY_UPSEG_TRAIN", "1") != "0"
# Add's backward hands the sum's gradient buffer to an operand that has no other consumer instead of copying it (Engine.add).
ADD_ALIAS = os.environ.get("DY_ADD_ALIAS", "1") != "0"
# The Add that follows ScalSeq (ASF models) folded into ScalSeq's tail kernel as a residual operand (nn/tasks.py, forward_act).
SCALSEQ_ADD = os.environ.get("DY_SCALSEQ_ADD", "1") != "0"
# Detect's final box convolution back-propagated from the ROWS of its output gradient (the loss writes box / DFL gradients for
# foreground anchors only): csrc/head_rows.hip reads the loss's assignment instead of a zero-filled dense gradient.  Only inside a
# StepPlan trace (the plan binds the assignment buffer and tells the loss not to zero the rest); DY_HEAD_ROWS=0: the dense kernels.
HEAD_ROWS = BIAS_WGRAD and os.environ.get("DY_HEAD_ROWS", "1") != "0"
# ... and its forward fused with the loss's box decode (dy_head_box_decode): inside a StepPlan trace the 64 fp32 logits per anchor are
# never written -- the assigner gets the decoded box, the loss recomputes the logits of foreground anchors.  DY_HEAD_DECODE=0: the
# conv writes its logits and the loss decodes them.
HEAD_DECODE = HEAD_ROWS and os.environ.get("DY_HEAD_DECODE", "1") != "0"
# ... and the Conv in front of it (cv2[l][1]) without an apply launch: its activated output is read by those kernels and by the loss's
# foreground-logit recompute only, all of which take (raw, coefficient table) and apply BatchNorm + SiLU where they load
# (common.h::bn_silu_apply8 -- the packed form of the apply kernel, hence only with DY_SILU_FAST on).  DY_HEAD_APPLY=0: apply launch.
# Detect's final class conv (nc <= 8) as stand-alone element-wise kernels, forward and one-walk backward (csrc/head_rows.hip,
# dy_cls_head_*), inside a StepPlan trace; with HEAD_APPLY the Conv in front leaves its apply out here too.  DY_HEAD_CLS=0: generic conv.
HEAD_CLS = BIAS_WGRAD and os.environ.get("DY_HEAD_CLS", "1") != "0"
# The head kernels of all detection levels in one launch per kind (dy_*_levels): the small levels run beside the large one.
HEAD_BATCH = os.environ.get("DY_HEAD_BATCH", "1") != "0"
# Independent Convs of one stage (Detect's six first convs, its six second convs) as a GROUP: their conv launches stay, their BatchNorm
# apply passes are one launch and so are their backward reduces (Engine.conv_bn_act_group, dy_bn_act_*_group).
BN_GROUP = BN_ACC and os.environ.get("DY_BN_GROUP", "1") != "0"
# Detect's levels as BRANCHES of the recorded step (Engine.branch): the 80x80 / 40x40 levels' conv stacks -- launches that cannot fill
# the chip -- are recorded on side streams and run beside the 160x160 level's, forward and backward (DY_HEAD_STREAMS=0: one stream).
HEAD_STREAMS = os.environ.get("DY_HEAD_STREAMS", "0") != "0"
# Fused (eval) models: Bottleneck's shortcut added in the epilogue of the conv in front of it (dy_conv_forward_res) instead of by a
# dy_add launch.  DY_CONV_RES=0: the launch pair.
CONV_RES = os.environ.get("DY_CONV_RES", "1") != "0"
# Concatenations that are never materialised (SegAct): every member of a C2f / Concat stays a contiguous tensor of its own and the 1x1
# conv behind the concat reads its input, writes its input gradient and reads its weight-gradient operand through a segment table
# (dy_conv1x1_*_segs).  Round 3 wrote the members into channel slices of one wide buffer: free to concatenate, but every kernel that
# touches ONE member (BatchNorm apply / backward reduce, the Bottleneck's 3x3 convs) then walked a strided slice and moved up to
# three times its bytes.  DY_PLANAR=0: the slices.
PLANAR = os.environ.get("DY_PLANAR", "1") != "0"
HEAD_APPLY = HEAD_DECODE and os.environ.get("DY_SILU_FAST", "1") != "0" and os.environ.get("DY_HEAD_APPLY", "1") != "0"
BN_DGRED = BN_WGRAD and os.environ.get("DY_BN_DGRED", "0") != "0"
BN_DGRED_MAXC = int(os.environ.get("DY_BN_DGRED_MAXC", "64"))
BN_DGRED_MAXPIX = int(os.environ.get("DY_BN_DGRED_MAXPIX", str(1 << 40)))  # ... and the map size (N*H*W): small maps are latency-bound


def dev_empty(shape, dtype, device):
    t = torch.empty(shape, dtype=dtype, device=device)
    if POISON and t.numel():
        t.view(-1).view(torch.uint8).fill_(0xFF)
    return t


class Arena:
    """One block of HBM that several recorded launch lists lay their step-local buffers over (activations, gradient twins,
    fp32 head logits, weight-gradient slabs, pool arg-max maps): every such buffer is written inside a step before it is read,
    and only one list runs at a time on the stream, so plans for different input sizes (``multi_scale`` training,
    engine/trainer.py) can share the footprint of the largest one instead of each holding its own.  Bump allocation, reset at the
    start of a trace; ``measure=True`` hands out ordinary allocations and only counts (used once, on the largest size)."""

    ALIGN = 256

    def __init__(self, nbytes, device, measure=False):
        self.device, self.measure = torch.device(device), measure
        self.buf = None if measure else dev_empty(int(nbytes), torch.uint8, self.device)
        self.cap, self.off, self.peak = int(nbytes), 0, 0

    def reset(self):
        self.off = 0
        if POISON and self.buf is not None:
            self.buf.fill_(0xFF)

    def take(self, shape, dtype):
        n = int(np.prod(shape)) * torch.empty((), dtype=dtype).element_size()
        start = (self.off + self.ALIGN - 1) // self.ALIGN * self.ALIGN
        self.off = start + n
        self.peak = max(self.peak, self.off)
        if self.measure:
            return dev_empty(shape, dtype, self.device)
        if self.off > self.cap:
            raise MemoryError(f"step-local buffers need more than the shared arena's {self.cap / 2**30:.2f} GiB "
                              f"(request of {n / 2**20:.1f} MiB at offset {start / 2**30:.2f} GiB): the arena is sized on the largest input size")
        return self.buf[start:start + n].view(dtype).view(shape)


class Storage:
    """(N,H,W,C) fp16 buffer with an optional gradient twin and a record of which channel ranges of the twin hold data."""

    def __init__(self, eng, N, H, W, C, dtype=torch.float16):
        assert C % 8 == 0, f"channel count {C} must be a multiple of 8"
        self.eng, self.N, self.H, self.W, self.C = eng, N, H, W, C
        self.buf = eng.transient((N, H, W, C), dtype)
        self.gbuf = None
        self.gwritten = []  # list of (c0, c1) already holding gradient

    def act(self, c0=0, C=None):
        return Act(self, c0, self.C - c0 if C is None else C)

    @classmethod
    def grad_only(cls, eng, N, H, W, C):
        """A Storage that only ever holds a GRADIENT (an UpAct's: the up-sampled tensor itself is never written)."""
        st = object.__new__(cls)
        st.eng, st.N, st.H, st.W, st.C = eng, N, H, W, C
        st.buf, st.gbuf, st.gwritten = None, None, []
        return st

    @classmethod
    def over(cls, eng, t):
        """Storage over an existing contiguous (N,H,W,C) tensor (a plan's static input)."""
        assert t.dim() == 4 and t.is_contiguous() and t.shape[3] % 8 == 0
        st = object.__new__(cls)
        st.eng, (st.N, st.H, st.W, st.C) = eng, t.shape
        st.buf, st.gbuf, st.gwritten = t, None, []
        return st


class Act:
    """Channel slice of a Storage; the unit every op consumes/produces."""
    __slots__ = ("st", "c0", "C", "needs_grad")

    def __init__(self, st, c0, C, needs_grad=True):
        assert c0 % 8 == 0 and C % 8 == 0 and c0 + C <= st.C
        self.st, self.c0, self.C, self.needs_grad = st, c0, C, needs_grad

    N = property(lambda s: s.st.N)
    H = property(lambda s: s.st.H)
    W = property(lambda s: s.st.W)
    ld = property(lambda s: s.st.C)
    npix = property(lambda s: s.st.N * s.st.H * s.st.W)

    @property
    def ptr(self):
        return self.st.buf.data_ptr() + self.c0 * self.st.buf.element_size()

    def sub(self, c0, C):
        return Act(self.st, self.c0 + c0, C, self.needs_grad)

    # ---- gradient twin -------------------------------------------------------------------------------------------
    def _gbuf(self):
        st = self.st
        if st.gbuf is None:
            st.gbuf = (st.eng.transient(tuple(st.buf.shape), st.buf.dtype) if st.buf is not None
                       else st.eng.transient((st.N, st.H, st.W, st.C), torch.float16))
            st.eng.hold(st.gbuf)
        return st.gbuf

    @property
    def gptr(self):
        g = self._gbuf()
        return g.data_ptr() + self.c0 * g.element_size()

    def grad_ready(self):
        """True when every channel of this slice has received gradient."""
        c0, c1 = self.c0, self.c0 + self.C
        cov = sorted(self.st.gwritten)
        pos = c0
        for a, b in cov:
            if a > pos:
                break
            pos = max(pos, b)
            if pos >= c1:
                return True
        return pos >= c1

    def grad_target(self):
        """-> accumulate flag for a writer of this slice's gradient; marks the range as written."""
        gw = self.st.eng._gw
        k = (id(self.st), self.c0, self.C)
        gw[k] = gw.get(k, 0) + 1
        c0, c1 = self.c0, self.c0 + self.C
        self._gbuf()
        overl = [(a, b) for a, b in self.st.gwritten if a < c1 and b > c0]
        if not overl:
            self.st.gwritten.append((c0, c1))
            return 0
        if self.grad_ready():
            return 1
        # partially covered: zero the uncovered granules, then accumulate
        covered = torch.zeros(self.st.C, dtype=torch.bool)
        for a, b in self.st.gwritten:
            covered[a:b] = True
        c = c0
        while c < c1:
            if not covered[c]:
                e = c
                while e < c1 and not covered[e]:
                    e += 1
                self.st.eng.zero_grad_slice(self.st, c, e)
                c = e
            else:
                c += 1
        self.st.gwritten.append((c0, c1))
        return 1


class ImageAct:
    """The image batch as the model's first operand WITHOUT an imported copy: a (N,3,H,W) fp32 tensor (x ``mul``) that only the direct
    stem path (Engine.conv_bn_act -> csrc/stem.hip) consumes; anything else asks for ``materialize()``."""
    needs_grad = False

    def __init__(self, eng, img, mul=1.0):
        assert img.dim() == 4 and img.shape[1] == 3 and img.dtype == torch.float32 and img.is_contiguous()
        self.eng, self.img, self.mul = eng, img, float(mul)
        self.N, self.C, self.H, self.W = img.shape[0], 3, img.shape[2], img.shape[3]
        self._act = None

    def materialize(self):
        if self._act is None:
            self._act = self.eng.import_image(self.img, 8, self.mul)
        return self._act


class UpAct:
    """``nn.Upsample(None, 2, 'nearest')`` of ``src`` that has not been executed.  As a member of a SegAct the segmented 1x1 conv reads
    the low-resolution tensor directly (forward and weight gradient); its input gradient goes to a full-resolution gradient tensor
    of this object's own, which Upsample's backward (the 2x2 sums) folds into ``src``'s.  Anything else asks ``Engine.dense`` to run
    the launch after all."""

    def __init__(self, src):
        self.src = src
        self.N, self.H, self.W, self.C = src.N, 2 * src.H, 2 * src.W, src.C
        self.st, self.c0 = src.st, src.c0  # (dtype checks look at the storage)
        self.needs_grad = src.needs_grad
        self._full = None
        self._g = None

    ld = property(lambda s: s.src.ld)
    ptr = property(lambda s: s.src.ptr)
    npix = property(lambda s: 4 * s.src.npix)

    @property
    def g(self):
        if self._g is None:
            self._g = Storage.grad_only(self.src.st.eng, self.N, self.H, self.W, self.C).act()
        return self._g

    gptr = property(lambda s: s.g.gptr)
    gld = property(lambda s: s.g.ld)

    def grad_target(self):
        return self.g.grad_target()

    def grad_ready(self):
        return self.g.grad_ready()


class SegAct:
    """A channel concatenation that is never materialised: ``parts`` are Acts (tensors or slices) of one (N, H, W), in channel order.
    Consumed by 1x1 convolutions through a segment table (``Engine._segs``); anything else asks ``Engine.dense`` for a copy."""

    def __init__(self, parts):
        flat = []
        for p in parts:
            flat.extend(p.parts if isinstance(p, SegAct) else [p])
        assert all((q.N, q.H, q.W) == (flat[0].N, flat[0].H, flat[0].W) for q in flat), "concatenation of different map sizes"
        self.parts = flat
        self.C = sum(q.C for q in flat)

    N = property(lambda s: s.parts[0].N)
    H = property(lambda s: s.parts[0].H)
    W = property(lambda s: s.parts[0].W)
    npix = property(lambda s: s.parts[0].npix)
    needs_grad = property(lambda s: any(q.needs_grad for q in s.parts))


class Recorder:
    def __init__(self):
        self.ops = []  # (cfunc, argtuple, name, stream id); cfunc None = fork/join marker (argtuple = (stream id,)); stream 0 = main


class Tape(list):
    """The backward closures of a trace.  A closure appended while the engine records on a BRANCH (``Engine.branch``: a side stream
    that runs an independent chain of the graph beside the main one) issues its backward launches on that branch's stream too."""

    def __init__(self, eng):
        super().__init__()
        self.eng = eng

    def append(self, f):
        sid = self.eng.cur_sid
        if sid:
            eng, g = self.eng, f
            f = lambda: eng._on(sid, g)  # noqa: E731
        super().append(f)


class ConvSpec:
    """Device-side view of one convolution (+ optional BatchNorm) of the model: master fp32 parameters (views into
    the flat parameter buffer), their gradient views, MFMA-packed fp16 weights and BN coefficient buffers."""

    def __init__(self, name, weight, bias, bn, ks, stride, act, bn_eps=BN2D_EPS, bn_mom=BN2D_MOM):
        self.name, self.weight, self.bias, self.bn = name, weight, bias, bn  # bn: dict(weight,bias,running_mean,running_var,nbt)|None
        self.ks, self.stride, self.act = ks, stride, act
        self.cout, self.cin = weight.shape[0], weight.shape[1]
        self.ld = None  # (N, C_phys, real_cin) for LDConv's (N,1) column conv seen as a 1x1 conv over N*C_phys channels
        self.bn_eps, self.bn_mom = bn_eps, bn_mom
        self.wpack = self.wpack_t = self.coef = self.bwdcoef = None
        self.acc_f = self.acc_b = None  # fp64 statistic accumulators [DY_BN_COPIES][2][cout] (forward sums, backward sums)
        self.acc_bias = None            # convs with bias: [DY_BN_COPIES][round8(cout)] sums of dY (bias gradient)
        self.gweight = self.gbias = self.gbn_w = self.gbn_b = None  # fp32 gradient views


class Engine:
    """Owns scratch memory, the tape, the recorder and the stream handle; all ops are methods."""

    def __init__(self, device):
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("the DEAL-YOLO HIP engine needs a GPU device (no CPU fallback exists by design)")
        self.L = lib()
        self.keep = []  # keep-alive list for buffers referenced by recorded launches
        self._tape = None  # list of backward closures while training-tracing (property ``tape``)
        self.rec = None  # active Recorder
        self._scratch = {}
        self._ident = {}
        self._gw = {}         # gradient writers per exact Act slice (grad_target calls) ...
        self._uses = {}       # ... forward consumers per storage {id(st): {(c0, C): count}} ...
        self._prod = {}       # ... and the Conv (spec, raw Act) that produced a slice: dataflow facts of the trace in progress that
        self._reduced = set() # decide where a BatchNorm backward reduce can ride on the dgrad that writes its gradient (BN_DGRED)
        self._unapplied = {}  # activations whose apply launch was left out: {(id(st), c0, C): (raw Act, spec)} (conv_bn_act, defer_apply)
        self._rows_grad = {}  # slices whose gradient was written by dy_conv1x1_rows_backward alone: {(id(st), c0, C): (assigned, A, a0)}
        self.training = False
        self._tmp_int = C.c_int(0)

    # ---- launch plumbing ------------------------------------------------------------------------------------------
    @property
    def stream(self):
        return torch.cuda.current_stream(self.device).cuda_stream

    def call(self, name, *args, side=False):
        """Launch (and, while tracing, record) one C-ABI call on the stream of the branch being recorded (``cur_sid``; 0 = the main
        stream).  ``side=True`` puts it on side stream 1: only for work nothing on the main stream depends on until the next
        ``join()`` (weight gradients), bracketed by ``fork()``."""
        fn = getattr(self.L, name)
        sid = 1 if side else self.cur_sid
        s = self._stream_of(sid).cuda_stream if sid else self.stream
        rc = fn(*args, s)
        check(rc, name)
        if self.rec is not None:
            self.rec.ops.append((fn, args, name, sid))

    # ---- two-stream plumbing: fork = "side waits for everything issued on main so far", join = "main waits for side" ----
    side_wgrad = False      # StepPlan switches it on for its backward trace
    _side_used = False      # a weight gradient of this backward pass went to the side stream (SIDE_SMALL): flush_wgrad joins
    cur_sid = 0             # stream id launches go to (0 = the caller's current stream); set by ``branch`` / ``_on``
    _side_streams = None
    _FORK, _JOIN = "<fork>", "<join>"

    def _stream_of(self, sid):
        if self._side_streams is None:
            self._side_streams = {}
        st = self._side_streams.get(sid)
        if st is None:
            st = self._side_streams[sid] = torch.cuda.Stream(self.device)
        return st

    @property
    def side_stream(self):
        return self._stream_of(1)

    def _sync(self, kind, sid=1):
        main, side = torch.cuda.current_stream(self.device), self._stream_of(sid)
        ev = torch.cuda.Event()
        if kind == self._FORK:
            ev.record(main)
            side.wait_event(ev)
        else:
            ev.record(side)
            main.wait_event(ev)

    def fork(self, sid=1):
        """Side stream ``sid`` waits for everything issued on the main stream so far."""
        self._sync(self._FORK, sid)
        if self.rec is not None:
            self.rec.ops.append((None, (sid,), self._FORK, 0))

    def join(self, sid=1):
        """The main stream waits for everything issued on side stream ``sid`` so far."""
        self._sync(self._JOIN, sid)
        if self.rec is not None:
            self.rec.ops.append((None, (sid,), self._JOIN, 0))

    # ---- branches: an independent chain of the graph (a detection level's head, ScalSeq beside the neck) recorded on its own stream,
    # so that its launches -- too small to fill the chip -- run beside the main chain's.  ``fork_branch`` / ``join_branch`` bracket it in
    # the forward AND leave the mirrored synchronisation on the tape: where the forward joined, the backward forks, and vice versa.
    def fork_branch(self, sid):
        assert self.cur_sid == 0
        self.fork(sid)
        if self.tape is not None:
            self.tape.append(lambda: self.join(sid))

    def join_branch(self, sid):
        assert self.cur_sid == 0
        self.join(sid)
        if self.tape is not None:
            self.tape.append(lambda: self.fork(sid))

    def branch(self, sid):
        eng = self

        class _Branch:
            def __enter__(self_b):
                self_b.prev, eng.cur_sid = eng.cur_sid, sid

            def __exit__(self_b, *exc):
                eng.cur_sid = self_b.prev
        return _Branch()

    def _on(self, sid, f):
        prev, self.cur_sid = self.cur_sid, sid
        try:
            f()
        finally:
            self.cur_sid = prev

    def replay(self, rec, lo=0, hi=None):
        """Re-issue a recorded launch list (or its slice [lo, hi)) on the CURRENT stream (which may be a capturing stream);
        side-stream launches and their fork/join points are reproduced (under capture they become parallel branches of the graph)."""
        s = self.stream
        for fn, args, name, sid in (rec.ops if lo == 0 and hi is None else rec.ops[lo:hi]):
            if fn is None:
                self._sync(name, args[0] if args else 1)
                continue
            rc = fn(*args, self._stream_of(sid).cuda_stream if sid else s)
            if rc != 0:
                check(rc, name)

    @property
    def tape(self):
        return self._tape

    @tape.setter
    def tape(self, v):
        if v is not None and not isinstance(v, Tape):
            t = Tape(self)
            list.extend(t, v)
            v = t
        self._tape = v
        if v is not None and len(v) == 0:  # a new trace begins: the dataflow facts of the previous one (keyed by object ids) are void
            self.reset_dataflow()

    def reset_dataflow(self):
        self._gw.clear(); self._uses.clear(); self._prod.clear(); self._reduced.clear(); self._rows_grad.clear(); self._unapplied.clear()

    def _use(self, *xs):
        """Forward bookkeeping: ``x`` will receive a gradient contribution from the op being traced."""
        if self.tape is None:
            return
        for x in xs:
            if isinstance(x, SegAct):
                self._use(*x.parts)
            elif isinstance(x, Act) and x.needs_grad:
                d = self._uses.setdefault(id(x.st), {})
                d[(x.c0, x.C)] = d.get((x.c0, x.C), 0) + 1

    def _sole_consumer_of_conv(self, x):
        """(spec, raw) of the Conv whose output IS ``x`` when the op now writing x's gradient is the only one that ever will: every
        forward use of x's storage was this one use of exactly this slice."""
        uses = self._uses.get(id(x.st))
        if uses is None or len(uses) != 1 or uses.get((x.c0, x.C)) != 1:
            return None
        return self._prod.get((id(x.st), x.c0, x.C))

    def hold(self, *ts):
        """Keep per-call buffers alive for recorded launches.  Eager (un-recorded) calls must NOT retain them: the caching
        allocator is stream-ordered, so a buffer freed by Python is only reused by launches enqueued later on this stream."""
        if self.rec is not None:
            self.keep.extend(ts)

    def scratch(self, key, nbytes):
        """Stream-ordered scratch reused across layers (partials, slabs, raw-gradient staging)."""
        t = self._scratch.get(key)
        if t is None or t.numel() < nbytes:
            # growing is safe while recording: launches already recorded keep using the old (kept-alive) buffer, and
            # scratch carries no state from one op to the next
            t = dev_empty(max(nbytes, 1 << 20), torch.uint8, self.device)
            self._scratch[key] = t
            self.keep.append(t)
        elif POISON:  # scratch carries no state between ops: what the previous user left is as undefined as fresh memory
            t.fill_(0xFF)
        return t

    # ---- fp64 statistic accumulators (BN_ACC): one pool for every BatchNorm of the model, zeroed by ONE launch at the start of a
    # recorded step (StepPlan) -- or slice by slice before use when a layer is run on its own
    acc_pool = None
    acc_used = 0
    acc_zeroed = False  # True while a trace runs whose launch list began with the pool memset

    def acc_take(self, cout, backward=False):
        n = DY_BN_COPIES * 2 * cout
        if self.acc_pool is None:
            self.acc_pool = torch.zeros(1 << 20, dtype=torch.float64, device=self.device)  # 8 MB: ~500 layers of 64 channels
            self.acc_bwd_mask = torch.zeros(1 << 20, dtype=torch.bool, device=self.device)
            self.keep.append(self.acc_pool)
        if self.acc_used + n > self.acc_pool.numel():
            raise MemoryError("BatchNorm accumulator pool exhausted")
        v = self.acc_pool[self.acc_used:self.acc_used + n]
        if backward:
            self.acc_bwd_mask[self.acc_used:self.acc_used + n] = True
        self.acc_used += n
        return v

    def zero_backward_acc(self):
        """The backward accumulators alone (eager): a backward half that runs on its own -- ``loss.backward()`` after
        ``model(batch)``, StepPlan.backward_accumulate -- must not add to sums an earlier backward pass left behind."""
        if self.acc_pool is not None:
            self.acc_pool.masked_fill_(self.acc_bwd_mask, 0.0)

    def zero_acc_pool(self):
        """Recorded as the first launch of a step: every accumulator of the model back to zero."""
        if self.acc_pool is not None and self.acc_used:
            self.call("dy_fill_zero", self.acc_pool.data_ptr(), self.acc_used * 8)
        self.acc_zeroed = True

    def _acc_ready(self, t):
        if not self.acc_zeroed:
            self.call("dy_fill_zero", t.data_ptr(), t.numel() * 8)
        return t.data_ptr()

    arena = None  # set by StepPlan around a trace whose step-local buffers go to a shared Arena

    def transient(self, shape, dtype):
        """A buffer that is written inside every step before it is read (never carries state from one step to the next)."""
        if self.arena is not None:
            return self.arena.take(tuple(shape), dtype)
        return dev_empty(tuple(shape), dtype, self.device)

    def new_storage(self, N, H, W, C, dtype=torch.float16):
        st = Storage(self, N, H, W, C, dtype)
        self.hold(st.buf)
        return st

    def new_act(self, N, H, W, C):
        return self.new_storage(N, H, W, C).act()

    def wrap_act(self, t):
        a = Storage.over(self, t).act()
        a.needs_grad = False
        return a

    def f32(self, n, fill=None):
        t = dev_empty(n, torch.float32, self.device) if fill is None else torch.full(
            (n,), float(fill), dtype=torch.float32, device=self.device)
        self.keep.append(t)
        return t

    def zero_grad_slice(self, st, c0, c1):
        # zero one channel range of a gradient twin: add(zeros) is not available, so use copy from a zero storage
        need = st.N * st.H * st.W * (c1 - c0) * 2
        z = getattr(self, "_zeros", None)
        if z is None or z.numel() < need:
            z = torch.zeros(need, dtype=torch.uint8, device=self.device)
            self._zeros = z
            self.keep.append(z)
        self.call("dy_copy_slice", z.data_ptr(), c1 - c0, st.gbuf.data_ptr() + 2 * c0, st.C, st.N * st.H * st.W, c1 - c0)

    # ---- parameter plumbing ---------------------------------------------------------------------------------------
    def prepare_conv(self, spec: ConvSpec):
        """Allocate packed-weight / coefficient buffers of a ConvSpec (idempotent)."""
        if spec.wpack is not None:
            return
        g = [C.c_int() for _ in range(8)]
        if spec.ld is not None:
            spec.cin = spec.ld[0] * spec.ld[1]
        cin_phys = (spec.cin + 7) // 8 * 8
        check(self.L.dy_conv_geometry(spec.cin, spec.cout, spec.ks, spec.stride, *[C.byref(x) for x in g]), "dy_conv_geometry")
        spec.wpack = torch.zeros(g[7].value, dtype=torch.float16, device=self.device)
        cout_phys = (spec.cout + 7) // 8 * 8
        check(self.L.dy_conv_geometry(cout_phys, spec.cin, spec.ks, 1, *[C.byref(x) for x in g]), "dy_conv_geometry")
        spec.wpack_t = torch.zeros(g[7].value, dtype=torch.float16, device=self.device)
        spec.cin_phys, spec.cout_phys = cin_phys, cout_phys
        if spec.bn is not None:
            spec.coef = self.f32(4 * spec.cout)
            spec.bwdcoef = self.f32(2 * spec.cout)
            if spec.cout % 16 == 0:
                spec.acc_f, spec.acc_b = self.acc_take(spec.cout), self.acc_take(spec.cout, backward=True)
        elif spec.bias is not None:
            cot = (spec.cout + 15) // 16
            if not (spec.ks == 1 and cot % 3 == 0 and cot % 4 != 0):  # 48- / 96-channel outputs: the three-tile kernel has no bias sums
                spec.acc_bias = self.acc_take((spec.cout + 7) // 8 * 8, backward=True)
        self.keep += [spec.wpack, spec.wpack_t]

    def pack(self, spec: ConvSpec, fold_scale=None, transposed=True):
        self.prepare_conv(spec)
        if spec.ld is not None:
            n, cphys, cin = spec.ld
            self.call("dy_pack_weights_ld", spec.weight.data_ptr(), spec.wpack.data_ptr(), spec.cout, cin, n, cphys, 0)
            if transposed:
                self.call("dy_pack_weights_ld", spec.weight.data_ptr(), spec.wpack_t.data_ptr(), spec.cout, cin, n, cphys, 1)
            return
        self.call("dy_pack_weights", spec.weight.data_ptr(), _ptr(fold_scale), spec.wpack.data_ptr(), spec.cout, spec.cin,
                  spec.ks, spec.stride, 0)
        if transposed:
            self.call("dy_pack_weights", spec.weight.data_ptr(), 0, spec.wpack_t.data_ptr(), spec.cout, spec.cin, spec.ks,
                      spec.stride, 1)

    def ident_coef(self, Cn):
        t = self._ident.get(Cn)
        if t is None:
            t = torch.cat([torch.ones(Cn), torch.zeros(Cn), torch.zeros(Cn), torch.ones(Cn)]).to(self.device)
            self._ident[Cn] = t
            self.keep.append(t)
        return t

    # ---- ops ------------------------------------------------------------------------------------------------------
    def import_image(self, img, cp=8, mul=1.0):
        """NCHW fp32 image batch -> NHWC fp16 Act with channels zero-padded to ``cp`` (no gradient)."""
        N, Cc, H, W = img.shape
        assert img.dtype == torch.float32 and img.is_contiguous()
        out = self.new_act(N, H, W, cp)
        out.needs_grad = False
        self.call("dy_import_image", img.data_ptr(), out.ptr, N, Cc, H, W, cp, float(mul))
        return out

    def import_image_u8(self, img, cp=8, flip=None, index=None, hsv=None):
        """NHWC uint8 RGB image batch (the loader's format) -> NHWC fp16 Act, value/255, channels zero-padded to ``cp``;
        ``flip``: (N,) uint8 device tensor of flip bits (1 = left-right, 2 = up-down) applied while converting;
        ``index``: (N,) int32 device tensor -- ``img`` then is a pool of images and slot i reads ``img[index[i]]``;
        ``hsv``: (N,3) float32 device tensor of RandomHSV gains."""
        N = img.shape[0] if index is None else index.shape[0]
        _, H, W, Cc = img.shape
        assert img.dtype == torch.uint8 and img.is_contiguous() and Cc == 3
        out = self.new_act(N, H, W, cp)
        out.needs_grad = False
        self.call("dy_import_image_u8", img.data_ptr(), out.ptr, N, H, W, cp, 0 if flip is None else flip.data_ptr(),
                  0 if index is None else index.data_ptr(), 0 if hsv is None else hsv.data_ptr())
        return out

    def import_warp(self, pool, slots, cp=8):
        """Mosaic + affine + flips composed from the HBM image pool into the fp16 NHWC stem input (dy_warp_import_u8);
        ``slots``: (N, 40) int32 device tensor of per-sample records (YOLODataset.warp_slot)."""
        assert pool.dtype == torch.uint8 and pool.is_contiguous() and pool.shape[1] == pool.shape[2] and pool.shape[3] == 3
        assert slots.dtype == torch.int32 and slots.is_contiguous() and slots.shape[1] * 4 == lib().dy_warp_slot_bytes()
        N, S = slots.shape[0], pool.shape[1]
        out = self.new_act(N, S, S, cp)
        out.needs_grad = False
        self.call("dy_warp_import_u8", pool.data_ptr(), slots.data_ptr(), out.ptr, N, S, cp)
        return out

    # ---- segmented concatenations -------------------------------------------------------------------------------------------------
    def _segs(self, x: SegAct, grad=False, acc=None):
        """The DySegs table of ``x`` (``grad``: of the members' gradient tensors, with their store / accumulate flags)."""
        t = DySegs()
        assert len(x.parts) <= 8
        t.nseg, end = len(x.parts), 0
        for i, q in enumerate(x.parts):
            end += q.C
            up = isinstance(q, UpAct)
            t.c_end[i], t.ld[i], t.ptr[i] = end, (q.gld if (grad and up) else q.ld), (q.gptr if grad else q.ptr)
            t.acc[i] = int(acc[i]) if acc is not None else (2 if (up and not grad) else 0)
        return t

    def seg_conv_ok(self, spec, x):
        """A 1x1 Conv can read the concatenation ``x`` without a copy: forward through dy_conv1x1_forward_segs and -- when a backward
        pass will follow -- weight / input gradients through their segmented forms (the BatchNorm-in-the-weight-gradient path)."""
        if not (PLANAR and isinstance(x, SegAct) and spec.ks == 1 and spec.stride == 1 and spec.ld is None and len(x.parts) <= 8
                and x.C == spec.cin_phys and all(q.st.buf.dtype == torch.float16 for q in x.parts)):
            return False
        if self.tape is not None and not (self.training and BN_ACC and BN_WGRAD and spec.acc_b is not None and spec.act == DY_ACT_SILU
                                          and not self.side_wgrad and spec.bn is not None):
            return False
        return bool(self.L.dy_conv1x1_segs_supported(x.C, spec.cout, C.byref(self._segs(x))))

    def snapshot(self, x):
        """``x`` as one tensor for INSPECTION (a per-layer capture): a copy that never joins the backward pass -- a copy made by
        ``dense`` registers a closure that hands ITS gradient on, and an inspection copy has no consumer to write one."""
        if not isinstance(x, (SegAct, UpAct)):
            return x
        tape, self._tape = self._tape, None  # (not through the property: an empty tape there means "a new trace begins")
        try:
            return self._upsample_now(x.src) if isinstance(x, UpAct) else self._concat_copy(x.parts)
        finally:
            self._tape = tape

    def planes_ok(self, spec, x):
        """``spec`` (a 1x1 Conv + BatchNorm + SiLU) may write its output as two planes: the training forms that take them are the
        accumulator-statistics apply, the split backward reduce and the BatchNorm-in-the-weight-gradient kernel."""
        if not (PLANAR and PLANAR_CV1 and self.training and BN_ACC and BN_WGRAD and spec.acc_f is not None and spec.acc_b is not None
                and spec.ks == 1 and spec.stride == 1 and spec.ld is None and spec.act == DY_ACT_SILU and spec.bn is not None
                and not self.side_wgrad and spec.cout % 16 == 0):
            return False
        return not isinstance(x, SegAct) or self.seg_conv_ok(spec, x)

    def new_planes(self, N, H, W, c):
        """Two (N, H, W, c) tensors in ONE allocation (and so their gradient twins): the halves of a C2f.cv1 output."""
        buf = self.transient((2, N, H, W, c), torch.float16)
        sts = [Storage.over(self, buf[0]), Storage.over(self, buf[1])]
        if self.tape is not None:
            g = self.transient((2, N, H, W, c), torch.float16)
            self.hold(g)
            sts[0].gbuf, sts[1].gbuf = g[0], g[1]
        self.hold(buf)
        return sts[0].act(), sts[1].act()

    def dense(self, x):
        """``x`` as ONE tensor: a SegAct is copied together (and its gradient split again in the backward pass), an UpAct is executed;
        Acts pass through."""
        if isinstance(x, UpAct):
            if x._full is None:
                x._full = self._upsample_now(x.src, optional=True)
            return x._full
        return self._concat_copy(x.parts) if isinstance(x, SegAct) else x

    def _conv_raw(self, spec, x, y_ptr, ldy, epi, partials_ptr=0, bias=None):
        if isinstance(x, SegAct):
            self.call("dy_conv1x1_forward_segs", C.byref(self._segs(x)), spec.wpack.data_ptr(), _ptr(bias), y_ptr, ldy, partials_ptr, x.N, x.H,
                      x.W, x.C, spec.cout, epi)
            return
        self.call("dy_conv_forward", x.ptr, x.ld, spec.wpack.data_ptr(), _ptr(bias), y_ptr, ldy, partials_ptr, x.N, x.H, x.W,
                  x.C, spec.cout, spec.ks, spec.stride, 1, 0, 0, epi, None)

    def out_hw(self, spec, x):
        p = spec.ks // 2
        return (x.H + 2 * p - spec.ks) // spec.stride + 1, (x.W + 2 * p - spec.ks) // spec.stride + 1

    def conv_bn_act(self, spec: ConvSpec, x: Act, out: Act | None = None, res: Act | None = None, defer_apply=False):
        """Conv.forward (reference nn/modules/conv.py:49-55) with training-mode BatchNorm; optional fused residual
        (Bottleneck.forward, nn/modules/block.py:333-335).  In eval mode BN uses running statistics.  ``defer_apply`` (training,
        accumulator path, no residual; the caller guarantees that every consumer of the result asks ``unapplied()`` for (raw, spec)
        instead of reading it): the apply launch only leaves the coefficient table and the running statistics behind (zero pixels),
        the returned activation is never written."""
        if isinstance(x, ImageAct):
            if (self.training and STEM_DIRECT and spec.acc_f is not None and (spec.cin, spec.cout, spec.ks, spec.stride) == (3, 16, 3, 2)
                    and spec.act == DY_ACT_SILU and res is None and spec.ld is None):
                return self._stem_direct(spec, x, out)
            if not self.training and self.tape is None and STEM_DIRECT and self._stem_eval_ok(spec, res):
                # eval, un-fused: the raw conv from the image batch itself, then BatchNorm with running statistics as for any Conv
                Ho, Wo = (x.H - 1) // 2 + 1, (x.W - 1) // 2 + 1
                raw = self.new_act(x.N, Ho, Wo, spec.cout)
                y = out if out is not None else self.new_act(x.N, Ho, Wo, spec.cout)
                bn = spec.bn
                self.call("dy_stem_forward_eval", x.img.data_ptr(), spec.weight.data_ptr(), 0, raw.ptr, raw.ld, x.N, x.H, x.W, x.mul, 0)
                self.call("dy_bn_eval_coef", bn["weight"].data_ptr(), bn["bias"].data_ptr(), bn["running_mean"].data_ptr(),
                          bn["running_var"].data_ptr(), spec.coef.data_ptr(), spec.cout, spec.bn_eps)
                self.call("dy_bn_act_apply", raw.ptr, raw.ld, 0, 0, y.ptr, y.ld, spec.coef.data_ptr(), y.npix, spec.cout, spec.act)
                return y
            x = x.materialize()
        if isinstance(x, UpAct) or (isinstance(x, SegAct) and not self.seg_conv_ok(spec, x)):
            x = self.dense(x)
        assert x.C == spec.cin_phys, (spec.name, x.C, spec.cin_phys)
        Ho, Wo = self.out_hw(spec, x)
        raw = self.new_act(x.N, Ho, Wo, spec.cout)
        y = out if out is not None else self.new_act(x.N, Ho, Wo, spec.cout)
        self._use(x, res)
        planes = isinstance(y, SegAct)  # the output as two planes (new_planes; the caller asked planes_ok)
        if self.tape is not None and not planes:
            self._prod[(id(y.st), y.c0, y.C)] = (spec, raw)
        assert (y.N, y.H, y.W, y.C) == (x.N, Ho, Wo, spec.cout), (spec.name, (y.N, y.H, y.W, y.C), (x.N, Ho, Wo, spec.cout))
        npix = x.N * Ho * Wo
        bn = spec.bn
        acc = self.training and BN_ACC and spec.acc_f is not None
        if planes:
            assert acc and res is None and not defer_apply and len(y.parts) == 2 and self.planes_ok(spec, x), spec.name
            y0, y1 = y.parts
            self._conv_raw(spec, x, raw.ptr, raw.ld, DY_EPI_STATS | DY_EPI_STATS_ACC, self._acc_ready(spec.acc_f))
            self.call("dy_bn_act_apply_acc_split", raw.ptr, raw.ld, y0.ptr, y0.ld, y1.ptr, y1.ld, y0.C, spec.acc_f.data_ptr(),
                      bn["weight"].data_ptr(), bn["bias"].data_ptr(), bn["running_mean"].data_ptr(), bn["running_var"].data_ptr(),
                      spec.coef.data_ptr(), npix, spec.cout, spec.act, float(npix), spec.bn_eps, spec.bn_mom)
            if self.tape is not None:
                self.tape.append(lambda: self._conv_bn_act_bwd(spec, x, raw, y, None))
            return y
        if acc:
            # statistics through the fp64 accumulator: conv adds, the apply kernel below finishes them in its prologue
            self._conv_raw(spec, x, raw.ptr, raw.ld, DY_EPI_STATS | DY_EPI_STATS_ACC, self._acc_ready(spec.acc_f))
            deferred = bool(defer_apply and res is None and spec.act == DY_ACT_SILU and self.tape is not None)
            self.call("dy_bn_act_apply_acc", raw.ptr, raw.ld, 0 if res is None else res.ptr, 0 if res is None else res.ld, y.ptr,
                      y.ld, spec.acc_f.data_ptr(), bn["weight"].data_ptr(), bn["bias"].data_ptr(), bn["running_mean"].data_ptr(),
                      bn["running_var"].data_ptr(), spec.coef.data_ptr(), 0 if deferred else npix, spec.cout, spec.act, float(npix),
                      spec.bn_eps, spec.bn_mom)
            if deferred:
                self._unapplied[(id(y.st), y.c0, y.C)] = (raw, spec)
        elif self.training:
            nparts = self.L.dy_conv_num_partials(x.N, x.H, x.W, x.C, spec.cout, spec.ks, spec.stride, 1)
            part = self.scratch("partials", nparts * 2 * ((spec.cout + 15) // 16 * 16) * 4 + 4096)
            self._conv_raw(spec, x, raw.ptr, raw.ld, DY_EPI_STATS, part.data_ptr())
            cp16 = (spec.cout + 15) // 16 * 16
            assert cp16 == spec.cout, "BN channel counts must be multiples of 16"
            self.call("dy_bn_finalize", part.data_ptr(), nparts, 1.0, 0, 0, 0.0, 0, 0, 0.0, bn["weight"].data_ptr(),
                      bn["bias"].data_ptr(), bn["running_mean"].data_ptr(), bn["running_var"].data_ptr(),
                      spec.coef.data_ptr(), spec.cout, float(npix), spec.bn_eps, spec.bn_mom, 1)
        else:
            self._conv_raw(spec, x, raw.ptr, raw.ld, 0)
            self.call("dy_bn_eval_coef", bn["weight"].data_ptr(), bn["bias"].data_ptr(), bn["running_mean"].data_ptr(),
                      bn["running_var"].data_ptr(), spec.coef.data_ptr(), spec.cout, spec.bn_eps)
        if not acc:
            self.call("dy_bn_act_apply", raw.ptr, raw.ld, 0 if res is None else res.ptr, 0 if res is None else res.ld, y.ptr, y.ld,
                      spec.coef.data_ptr(), npix, spec.cout, spec.act)
        if self.tape is not None:
            self.tape.append(lambda: self._conv_bn_act_bwd(spec, x, raw, y, res))
        return y

    @staticmethod
    def _stem_eval_ok(spec, res):
        return (spec.cin, spec.cout, spec.ks, spec.stride) == (3, 16, 3, 2) and res is None and spec.ld is None

    def _stem_direct(self, spec, x: ImageAct, out=None):
        """model.0 from the image batch itself (csrc/stem.hip): forward conv + statistics, the usual apply; backward = the usual
        reduce, then the weight gradient with the BatchNorm / SiLU backward apply inside (no input gradient exists)."""
        Ho, Wo = (x.H - 1) // 2 + 1, (x.W - 1) // 2 + 1
        bn = spec.bn
        raw = self.new_act(x.N, Ho, Wo, spec.cout)
        y = out if out is not None else self.new_act(x.N, Ho, Wo, spec.cout)
        npix = x.N * Ho * Wo
        self.call("dy_stem_forward", x.img.data_ptr(), spec.weight.data_ptr(), raw.ptr, raw.ld, self._acc_ready(spec.acc_f), x.N, x.H, x.W, x.mul)
        self.call("dy_bn_act_apply_acc", raw.ptr, raw.ld, 0, 0, y.ptr, y.ld, spec.acc_f.data_ptr(), bn["weight"].data_ptr(),
                  bn["bias"].data_ptr(), bn["running_mean"].data_ptr(), bn["running_var"].data_ptr(), spec.coef.data_ptr(), npix, spec.cout,
                  spec.act, float(npix), spec.bn_eps, spec.bn_mom)
        if self.tape is not None:
            def bwd():
                assert y.grad_ready(), f"gradient of {spec.name} output incomplete"
                self.call("dy_bn_act_bwd_reduce_acc", y.gptr, y.ld, raw.ptr, raw.ld, spec.coef.data_ptr(), self._acc_ready(spec.acc_b), npix,
                          spec.cout, spec.act, 0, 0, 0)
                ns = self.L.dy_stem_grid(x.N, x.H, x.W)
                se = 9 * 16 * 16
                deferred = self.deferred_wgrad is not None
                slabs = self.transient((ns * se,), torch.float32) if deferred else self.scratch("slabs", ns * se * 4)
                self.hold(slabs)
                self.call("dy_stem_wgrad_bn", x.img.data_ptr(), y.gptr, y.ld, raw.ptr, raw.ld, spec.coef.data_ptr(), spec.acc_b.data_ptr(),
                          spec.gbn_w.data_ptr(), spec.gbn_b.data_ptr(), float(npix), slabs.data_ptr(), x.N, x.H, x.W, x.mul)
                if deferred:
                    self.deferred_wgrad.append((spec, slabs, ns))
                else:  # reduce now (one descriptor through the batched entry)
                    keep, self.deferred_wgrad = self.deferred_wgrad, [(spec, slabs, ns)]
                    self.flush_wgrad()
                    self.deferred_wgrad = keep
            self.tape.append(bwd)
        return y

    def conv_bn_act_group(self, members, defer=None):
        """``conv_bn_act`` for independent Convs of one stage: members = [(spec, x)] (training, accumulator path, SiLU, no residual, own
        outputs); ``defer[i]``: member i leaves its apply out (``defer_apply``).  The conv launches are issued one after the other, then
        ONE dy_bn_act_apply_acc_group; the backward is ONE closure: the dense backward reduces of the members in one launch (a member whose
        gradient has rows keeps its own rows reduce), then each member's weight / input gradient."""
        defer = list(defer) if defer is not None else [False] * len(members)
        ok = (BN_GROUP and self.training and self.tape is not None and 1 < len(members) <= self.L.dy_bn_group_max()
              and all(sp.acc_f is not None and sp.acc_b is not None and sp.act == DY_ACT_SILU and sp.ld is None and not isinstance(x, ImageAct)
                      for sp, x in members))
        if not ok:
            return [self.conv_bn_act(sp, x, defer_apply=d) for (sp, x), d in zip(members, defer)]
        outs = []
        for (spec, x), d in zip(members, defer):
            assert x.C == spec.cin_phys, (spec.name, x.C, spec.cin_phys)
            Ho, Wo = self.out_hw(spec, x)
            raw = self.new_act(x.N, Ho, Wo, spec.cout)
            y = self.new_act(x.N, Ho, Wo, spec.cout)
            self._use(x)
            self._prod[(id(y.st), y.c0, y.C)] = (spec, raw)
            self._conv_raw(spec, x, raw.ptr, raw.ld, DY_EPI_STATS | DY_EPI_STATS_ACC, self._acc_ready(spec.acc_f))
            outs.append((spec, x, raw, y, bool(d)))
        P, I, Lg, F = C.c_void_p, C.c_int, C.c_long, C.c_float
        bn = [sp.bn for sp, *_ in outs]
        npx = [y.npix for _, _, _, y, _ in outs]
        self.call("dy_bn_act_apply_acc_group", len(outs), self._arr(P, [raw.ptr for _, _, raw, _, _ in outs]), self._arr(I, [raw.ld for _, _, raw, _, _ in outs]),
                  self._arr(P, [y.ptr for _, _, _, y, _ in outs]), self._arr(I, [y.ld for _, _, _, y, _ in outs]),
                  self._arr(P, [sp.acc_f.data_ptr() for sp, *_ in outs]), self._arr(P, [b["weight"].data_ptr() for b in bn]),
                  self._arr(P, [b["bias"].data_ptr() for b in bn]), self._arr(P, [b["running_mean"].data_ptr() for b in bn]),
                  self._arr(P, [b["running_var"].data_ptr() for b in bn]), self._arr(P, [sp.coef.data_ptr() for sp, *_ in outs]),
                  self._arr(Lg, [0 if d else n for n, (*_, d) in zip(npx, outs)]), self._arr(I, [sp.cout for sp, *_ in outs]),
                  self._arr(F, [float(n) for n in npx]), self._arr(F, [sp.bn_eps for sp, *_ in outs]), self._arr(F, [sp.bn_mom for sp, *_ in outs]))
        for spec, x, raw, y, d in outs:
            if d:
                self._unapplied[(id(y.st), y.c0, y.C)] = (raw, spec)

        def bwd():
            dense = []
            for spec, x, raw, y, _ in outs:
                assert y.grad_ready(), f"gradient of {spec.name} output incomplete"
                ykey = (id(y.st), y.c0, y.C)
                rows = self._rows_grad.get(ykey)
                if rows is not None and self._gw.get(ykey) == 1:
                    asg, A, a0 = rows
                    self.call("dy_bn_act_bwd_reduce_rows", y.gptr, y.ld, raw.ptr, raw.ld, spec.coef.data_ptr(), self._acc_ready(spec.acc_b),
                              y.N, y.H * y.W, spec.cout, spec.act, asg, A, a0)
                else:
                    dense.append((spec, raw, y))
            if len(dense) == 1:
                spec, raw, y = dense[0]
                self.call("dy_bn_act_bwd_reduce_acc", y.gptr, y.ld, raw.ptr, raw.ld, spec.coef.data_ptr(), self._acc_ready(spec.acc_b), y.npix,
                          spec.cout, spec.act, 0, 0, 0)
            elif dense:
                self.call("dy_bn_act_bwd_reduce_acc_group", len(dense), self._arr(P, [y.gptr for _, _, y in dense]), self._arr(I, [y.ld for _, _, y in dense]),
                          self._arr(P, [raw.ptr for _, raw, _ in dense]), self._arr(I, [raw.ld for _, raw, _ in dense]),
                          self._arr(P, [sp.coef.data_ptr() for sp, _, _ in dense]), self._arr(P, [self._acc_ready(sp.acc_b) for sp, _, _ in dense]),
                          self._arr(Lg, [y.npix for _, _, y in dense]), self._arr(I, [sp.cout for sp, _, _ in dense]))
            for spec, x, raw, y, _ in reversed(outs):
                self._conv_bn_act_bwd(spec, x, raw, y, None, reduced=True)
        self.tape.append(bwd)
        return [y for _, _, _, y, _ in outs]

    def _conv_bn_act_bwd(self, spec, x, raw, y, res, reduced=False):
        if isinstance(y, SegAct):  # two output planes (conv_bn_act): the split reduce, then the weight gradient with dY in planes
            y0, y1 = y.parts
            assert y0.grad_ready() and y1.grad_ready(), f"gradient of {spec.name} output incomplete"
            assert y1.gptr - y0.gptr == y0.npix * y0.ld * 2 and y0.ld == y1.ld == y0.C, "the planes' gradients are not one allocation"
            npix = y0.npix
            self.call("dy_bn_act_bwd_reduce_acc_split", y0.gptr, y0.ld, y1.gptr, y1.ld, y0.C, raw.ptr, raw.ld, spec.coef.data_ptr(),
                      self._acc_ready(spec.acc_b), npix, spec.cout, spec.act)
            if self.cur_sid:
                draw = self.transient((npix * spec.cout,), torch.float16)
                self.hold(draw)
            else:
                draw = self.scratch("draw", npix * spec.cout * 2)
            bn = (raw, draw, spec.coef.data_ptr(), spec.acc_b.data_ptr(), spec.gbn_w.data_ptr(), spec.gbn_b.data_ptr(), float(npix))
            self._conv_bwd(spec, x, y0.gptr, y0.ld, y0.H, y0.W, bn=bn, planes=(y1.gptr, y0.C))
            return
        assert y.grad_ready(), f"gradient of {spec.name} output incomplete"
        npix = y.npix
        acc = BN_ACC and spec.acc_b is not None
        fused_red = id(spec) in self._reduced  # the dgrad that wrote y's gradient already summed g and g*xhat into acc_b
        if fused_red and self._gw.get((id(y.st), y.c0, y.C)) != 1:
            raise RuntimeError(f"{spec.name}: BatchNorm backward reduce was fused into a dgrad that is not the only writer of the gradient")
        rg = (0, 0, 0)
        if res is not None and res.needs_grad:
            racc = res.grad_target()
            if acc and BN_RES and res.C == spec.cout and not fused_red:  # the shortcut's gradient (= dy) rides on the reduce pass below
                rg = (res.gptr, res.ld, racc)
            elif racc:
                self.call("dy_add", res.gptr, res.ld, y.gptr, y.ld, 0, 0, res.gptr, res.ld, npix, res.C)
            else:
                self.call("dy_copy_slice", y.gptr, y.ld, res.gptr, res.ld, npix, res.C)
        ykey = (id(y.st), y.c0, y.C)
        rows = self._rows_grad.get(ykey)
        if rows is not None and self._gw.get(ykey) != 1:
            rows = None  # somebody else wrote into that gradient as well: it is dense
        if acc and (fused_red or reduced):  # reduced: conv_bn_act_group ran this layer's reduce with its group's
            pass
        elif acc and rows is not None and rg == (0, 0, 0):
            # y's gradient came from dy_conv1x1_rows_backward alone: zero outside the loss's foreground anchors, so are the summands
            asg, A, a0 = rows
            self.call("dy_bn_act_bwd_reduce_rows", y.gptr, y.ld, raw.ptr, raw.ld, spec.coef.data_ptr(), self._acc_ready(spec.acc_b),
                      y.N, y.H * y.W, spec.cout, spec.act, asg, A, a0)
        elif acc:
            self.call("dy_bn_act_bwd_reduce_acc", y.gptr, y.ld, raw.ptr, raw.ld, spec.coef.data_ptr(), self._acc_ready(spec.acc_b),
                      npix, spec.cout, spec.act, *rg)
        else:
            part = self.scratch("partials", 2048 * 2 * spec.cout * 4 + 4096)
            n = C.c_int(0)
            self.call("dy_bn_act_bwd_reduce", y.gptr, y.ld, raw.ptr, raw.ld, spec.coef.data_ptr(), part.data_ptr(), 2048, npix,
                      spec.cout, spec.act, C.byref(n))
            self.call("dy_bn_bwd_finalize", part.data_ptr(), n.value, spec.gbn_w.data_ptr(), spec.gbn_b.data_ptr(),
                      spec.bwdcoef.data_ptr(), spec.cout, float(npix), 0)
        side_small = bool(SIDE_SMALL and npix <= SIDE_SMALL and self.deferred_wgrad is not None and not self.cur_sid and not self.side_wgrad
                          and acc and isinstance(x, Act) and x.needs_grad and spec.ld is None)
        if ((self.side_wgrad or side_small) and self.deferred_wgrad is not None) or self.cur_sid:
            # the weight gradient reads this buffer on the side stream while the main stream moves on to the next layer (or this
            # layer's backward runs on a branch beside the main chain's): it cannot be the shared scratch
            draw = self.transient((npix * spec.cout,), torch.float16)
            self.hold(draw)
        else:
            draw = self.scratch("draw", npix * spec.cout * 2)
        if acc and BN_WGRAD and spec.act == DY_ACT_SILU and not self.side_wgrad and not side_small and raw.ld == spec.cout:
            # no apply launch: the weight-gradient kernel forms d(raw) while staging and leaves it in ``draw`` for the dgrad
            bn = (raw, draw, spec.coef.data_ptr(), spec.acc_b.data_ptr(), spec.gbn_w.data_ptr(), spec.gbn_b.data_ptr(), float(npix))
            self._conv_bwd(spec, x, y.gptr, y.ld, y.H, y.W, bn=bn)
            return
        if acc:
            self.call("dy_bn_act_bwd_apply_acc", y.gptr, y.ld, raw.ptr, raw.ld, draw.data_ptr(), spec.cout, spec.coef.data_ptr(),
                      spec.acc_b.data_ptr(), spec.gbn_w.data_ptr(), spec.gbn_b.data_ptr(), npix, spec.cout, spec.act, float(npix))
        else:
            self.call("dy_bn_act_bwd_apply", y.gptr, y.ld, raw.ptr, raw.ld, draw.data_ptr(), spec.cout, spec.coef.data_ptr(),
                      spec.bwdcoef.data_ptr(), npix, spec.cout, spec.act, 0)
        self._conv_bwd(spec, x, draw.data_ptr(), spec.cout, y.H, y.W, side_hint=side_small)

    def _conv_bwd(self, spec, x, dy_ptr, lddy, Ho, Wo, accumulate_w=0, defer=True, bn=None, bias_acc=None, planes=None, side_hint=False):
        """weight gradient + input gradient of one convolution given d(raw output) (fp16, (N,Ho,Wo,lddy)).
        When the engine is collecting (``self.deferred_wgrad`` is a list: StepPlan's backward trace) the per-workgroup slabs of
        this layer are kept and reduced together with every other layer's by ONE ``dy_wgrad_reduce_batched`` launch at the end
        of the backward pass; weights shared by several calls (ScalSeq's conv3d: ``defer=False``) reduce immediately."""
        ns, se = C.c_int(), C.c_long()
        self.L.dy_wgrad_workspace(x.N, x.H, x.W, spec.cin, spec.cout, spec.ks, spec.stride, C.byref(ns), C.byref(se))
        deferred = defer and not accumulate_w and self.deferred_wgrad is not None
        if deferred:
            slabs = self.transient((ns.value * se.value,), torch.float32)
            self.hold(slabs)
            self.deferred_wgrad.append((spec, slabs, ns.value) if bias_acc is None else (spec, slabs, ns.value, bias_acc))
            dw = 0
        else:
            slabs = self.scratch("slabs", ns.value * se.value * 4)
            dw = spec.gweight.data_ptr()
        # deferred weight gradients are off the critical path (needed only by the batched reduction at the end of the
        # backward pass): with ``side_wgrad`` they run on the side stream beside the input-gradient chain
        side = deferred and (self.side_wgrad or side_hint)
        if side:
            self._side_used = True
            self.fork()
        if planes is not None:  # dY in two planes (dy_ptr, planes[0]), split at channel planes[1]; x a tensor or a concatenation
            raw, draw, coef, accb, gw, gb, cnt = bn
            seg = isinstance(x, SegAct)
            self.call("dy_conv1x1_wgrad_bn_planes", C.byref(self._segs(x)) if seg else None, 0 if seg else x.ptr, 0 if seg else x.ld, dy_ptr,
                      planes[0], lddy, planes[1], raw.ptr, raw.ld, draw.data_ptr() if x.needs_grad else 0, coef, accb, gw, gb, cnt,
                      slabs.data_ptr(), dw, x.N, x.H, x.W, spec.cin, spec.cout, accumulate_w)
            dy_ptr, lddy = draw.data_ptr(), spec.cout
        elif bn is not None and isinstance(x, SegAct):  # (seg_conv_ok checked at forward time that this path is the one taken)
            raw, draw, coef, accb, gw, gb, cnt = bn
            self.call("dy_conv1x1_wgrad_bn_segs", C.byref(self._segs(x)), dy_ptr, lddy, raw.ptr, raw.ld, draw.data_ptr() if x.needs_grad else 0,
                      coef, accb, gw, gb, cnt, slabs.data_ptr(), dw, x.N, x.H, x.W, spec.cin, spec.cout, accumulate_w)
            dy_ptr, lddy = draw.data_ptr(), spec.cout
        elif bn is not None:  # dy_ptr is the gradient w.r.t. the ACTIVATED output: BatchNorm + SiLU backward inside the kernel
            raw, draw, coef, accb, gw, gb, cnt = bn
            head = (x.ptr, x.ld, dy_ptr, lddy, raw.ptr, raw.ld, draw.data_ptr() if x.needs_grad else 0, coef, accb, gw, gb, cnt,
                    slabs.data_ptr(), dw, x.N, x.H, x.W)
            if spec.ld is not None:
                n, cphys, cin = spec.ld
                self.call("dy_conv_wgrad_ld_bn", *head, spec.cout, cin, n, cphys, accumulate_w)
            else:
                self.call("dy_conv_wgrad_bn", *head, spec.cin, spec.cout, spec.ks, spec.stride, accumulate_w)
            dy_ptr, lddy = draw.data_ptr(), spec.cout
        elif bias_acc is not None:
            assert deferred
            self.call("dy_conv_wgrad_bias", x.ptr, x.ld, dy_ptr, lddy, self._acc_ready(bias_acc), slabs.data_ptr(), dw, x.N, x.H, x.W,
                      spec.cin, spec.cout, spec.ks, spec.stride, accumulate_w)
        elif spec.ld is not None:
            n, cphys, cin = spec.ld
            self.call("dy_conv_wgrad_ld", x.ptr, x.ld, dy_ptr, lddy, slabs.data_ptr(), dw, x.N, x.H, x.W,
                      spec.cout, cin, n, cphys, accumulate_w, side=side)
        else:
            self.call("dy_conv_wgrad", x.ptr, x.ld, dy_ptr, lddy, slabs.data_ptr(), dw, x.N, x.H, x.W,
                      spec.cin, spec.cout, spec.ks, spec.stride, accumulate_w, side=side)
        if isinstance(x, SegAct):
            assert bn is not None, f"{spec.name}: a segmented input needs the BatchNorm-in-the-weight-gradient path"
            if x.needs_grad:  # every 8-channel piece of W^T d(raw) goes to its member's gradient tensor, stored or added per member
                accs = [q.grad_target() if q.needs_grad else 0 for q in x.parts]
                if not all(q.needs_grad for q in x.parts):  # a member without a gradient still receives its pieces: give it a sink
                    raise NotImplementedError("a concatenation member that needs no gradient beside members that do")
                self.call("dy_conv1x1_input_grad_segs", dy_ptr, lddy, spec.wpack_t.data_ptr(), C.byref(self._segs(x, grad=True, acc=accs)),
                          x.N, Ho, Wo, spec.cout_phys, spec.cin)
            return
        if x.needs_grad:
            acc = x.grad_target()
            prod = self._sole_consumer_of_conv(x) if (BN_DGRED and not acc and spec.stride == 1 and spec.ld is None) else None
            if prod is not None:
                ps, praw = prod
                if not (ps.act == DY_ACT_SILU and ps.acc_b is not None and ps.cout == x.C == spec.cin and x.C <= BN_DGRED_MAXC and x.npix <= BN_DGRED_MAXPIX
                        and (praw.N, praw.H, praw.W) == (x.N, x.H, x.W)
                        and self.L.dy_conv_red_supported(spec.cout_phys, spec.cin, spec.ks)):
                    prod = None
            if prod is not None:
                # this dgrad is the only writer of the gradient of ps's output: ps's BatchNorm backward reduce rides on its epilogue
                self.call("dy_conv_input_grad_red", dy_ptr, lddy, spec.wpack_t.data_ptr(), x.gptr, x.ld, x.N, Ho, Wo, spec.cout_phys, spec.cin,
                          spec.ks, praw.ptr, praw.ld, ps.coef.data_ptr(), self._acc_ready(ps.acc_b), ps.cout)
                self._reduced.add(id(ps))
            else:
                self.call("dy_conv_forward", dy_ptr, lddy, spec.wpack_t.data_ptr(), 0, x.gptr, x.ld, 0, x.N, Ho, Wo,
                          spec.cout_phys, spec.cin, spec.ks, 1, spec.stride, x.H, x.W, DY_EPI_ACCUM if acc else 0, None)

    deferred_wgrad = None
    tape_mark = None  # index into the tape where the neck + head begin (set by BaseModel.forward_act while tracing)

    def flush_wgrad(self):
        """Reduce the slabs of every deferred layer into its fp32 weight gradient: one launch, one descriptor per layer."""
        items, self.deferred_wgrad = self.deferred_wgrad, None
        if self.side_wgrad or self._side_used:
            self.join()
            self._side_used = False
        if not items:
            return
        sz = self.L.dy_wgrad_reduce_desc_bytes()
        host = (C.c_char * (sz * len(items)))()
        blocks = 0
        for i, (spec, slabs, ns, *bias) in enumerate(items):
            ld = spec.ld or (0, 0, 0)
            nb = self.L.dy_wgrad_reduce_desc_fill(C.byref(host, i * sz), slabs.data_ptr(), ns, spec.gweight.data_ptr(), spec.cin, spec.cout,
                                                  spec.ks, spec.stride, 0, ld[0], ld[1], ld[2], blocks)
            if nb < 0:
                raise RuntimeError(f"dy_wgrad_reduce_desc_fill failed for {spec.name}")
            if bias:  # this layer's weight-gradient kernel also summed dY: the reduction launch writes the bias gradient
                check(self.L.dy_wgrad_reduce_desc_bias(C.byref(host, i * sz), bias[0].data_ptr(), spec.gbias.data_ptr(), (spec.cout + 7) // 8 * 8),
                      "dy_wgrad_reduce_desc_bias")
            blocks += nb
        dev = torch.frombuffer(bytearray(host), dtype=torch.uint8).to(self.device)
        self.keep.append(dev)
        self.call("dy_wgrad_reduce_batched", dev.data_ptr(), len(items), blocks)

    def conv_fused(self, spec: ConvSpec, x: Act, out: Act | None = None, res: Act | None = None):
        """Conv.forward_fuse (reference nn/modules/conv.py:57-59): BN folded into weights + bias, SiLU in the epilogue."""
        if isinstance(x, ImageAct):
            if STEM_DIRECT and self._stem_eval_ok(spec, res) and spec.bias is not None:  # the fused stem from the image batch itself
                Ho, Wo = (x.H - 1) // 2 + 1, (x.W - 1) // 2 + 1
                y = out if out is not None else self.new_act(x.N, Ho, Wo, spec.cout)
                self.call("dy_stem_forward_eval", x.img.data_ptr(), spec.weight.data_ptr(), spec.bias.data_ptr(), y.ptr, y.ld, x.N, x.H, x.W,
                          x.mul, 1 if spec.act == DY_ACT_SILU else 0)
                return y
            x = x.materialize()
        if isinstance(x, UpAct) or (isinstance(x, SegAct) and not self.seg_conv_ok(spec, x)):
            x = self.dense(x)
        Ho, Wo = self.out_hw(spec, x)
        y = out if out is not None else self.new_act(x.N, Ho, Wo, spec.cout)
        epi = DY_EPI_BIAS | (DY_EPI_SILU if spec.act == DY_ACT_SILU else 0)
        if (res is not None and CONV_RES and spec.act == DY_ACT_SILU and spec.ld is None
                and self.L.dy_conv_res_supported(x.C, spec.cout, spec.ks, spec.stride)):
            # Bottleneck's shortcut rides on the conv's epilogue (same bits as the store + dy_add pair)
            self.call("dy_conv_forward_res", x.ptr, x.ld, spec.wpack.data_ptr(), spec.bias.data_ptr(), res.ptr, res.ld, y.ptr, y.ld,
                      x.N, x.H, x.W, x.C, spec.cout, spec.ks, spec.stride)
            return y
        self._conv_raw(spec, x, y.ptr, y.ld, epi, 0, spec.bias)
        if res is not None:
            self.call("dy_add", y.ptr, y.ld, res.ptr, res.ld, 0, 0, y.ptr, y.ld, y.npix, y.C)
        return y

    def conv_bias(self, spec: ConvSpec, x: Act, y_ptr, ldy, f32out=True, dy_ptr_fn=None, out_hw=None, rows_level=None):
        """Plain conv + bias (Detect's final nn.Conv2d 1x1, reference nn/modules/head.py:38-42).  ``dy_ptr_fn`` returns
        (ptr, ld) of the fp16 gradient w.r.t. the output at backward time.  ``rows_level``: this is the box branch of detection
        level ``rows_level`` -- its output gradient is non-zero at foreground anchors only (see HEAD_ROWS)."""
        x = self.dense(x)
        self._use(x)
        self._conv_raw(spec, x, y_ptr, ldy, DY_EPI_BIAS | (DY_EPI_F32OUT if f32out else 0), 0, spec.bias)
        if self.tape is not None:
            if (rows_level is not None and self.rows_used is not None and spec.ks == 1 and spec.ld is None and spec.acc_bias is not None
                    and self.L.dy_conv1x1_rows_supported(spec.cin, spec.cout)):
                self.rows_used.add(rows_level)
            else:
                rows_level = None
            self.tape.append(lambda: self._conv_bias_bwd(spec, x, dy_ptr_fn, rows_level=rows_level))

    def rows_capable(self, spec):
        return (self.rows_used is not None and self.tape is not None and spec.ks == 1 and spec.ld is None and spec.acc_bias is not None
                and spec.bias is not None and bool(self.L.dy_conv1x1_rows_supported(spec.cin, spec.cout)))

    def cls_capable(self, spec, ncp):
        return (HEAD_CLS and self.pending_decode is not None and self.tape is not None and spec.ks == 1 and spec.ld is None and ncp == 8
                and spec.acc_bias is not None and spec.bias is not None and bool(self.L.dy_cls_head_supported(spec.cin, spec.cout)))

    def conv_bias_cls(self, spec: ConvSpec, x: Act, y_ptr, dy_ptr_fn):
        """Detect's final class conv inside a StepPlan trace (caller checked ``cls_capable``): dy_cls_head_forward now, one
        dy_cls_head_backward launch for the weight, bias and input gradients later."""
        self._use(x)
        src = self.unapplied(x)
        xp, xld, xcoef = (x.ptr, x.ld, 0) if src is None else (src[0].ptr, src[0].ld, src[1].coef.data_ptr())
        self.call("dy_cls_head_forward", xp, xld, xcoef, spec.weight.data_ptr(), spec.bias.data_ptr(), y_ptr, x.npix, spec.cin, spec.cout)

        def bwd():
            dyp, ld = dy_ptr_fn()
            assert ld == 8 and self.deferred_wgrad is not None and not self.side_wgrad
            ns = self.L.dy_cls_head_slabs()
            slabs = self.transient((ns * 16 * spec.cin,), torch.float32)
            self.hold(slabs)
            self.deferred_wgrad.append((spec, slabs, ns, spec.acc_bias))
            acc = x.grad_target() if x.needs_grad else 0
            self.call("dy_cls_head_backward", xp, xld, xcoef, dyp, spec.weight.data_ptr(), x.gptr if x.needs_grad else 0, x.ld, acc,
                      slabs.data_ptr(), self._acc_ready(spec.acc_bias), x.npix, spec.cin, spec.cout)
        self.tape.append(bwd)

    @staticmethod
    def _arr(ctype, vals):
        return (ctype * len(vals))(*vals)

    def _src(self, x):
        """(pointer, ld, coefficient-table pointer or 0) a head kernel reads ``x`` through: its RAW twin when the apply was left out."""
        src = self.unapplied(x)
        return (x.ptr, x.ld, 0) if src is None else (src[0].ptr, src[0].ld, src[1].coef.data_ptr())

    def head_box_levels(self, items, dy_ptr_fns):
        """``conv_bias_decode`` for every detection level at once (items = [(spec, x, level)], all levels): the plan launches ONE
        dy_head_box_decode_levels, the backward is ONE dy_conv1x1_rows_backward_levels -- registered after every level's forward, so it
        runs first in the backward pass (it needs the loss's gradients only)."""
        for spec, x, l in items:
            self._use(x)
            self.rows_used.add(l)
            self.pending_decode.append((spec, x, l))
        self.tape.append(lambda: self._rows_bwd_levels(items, dy_ptr_fns))

    def _rows_bwd_levels(self, items, fns):
        if not (self.loss_rows is not None and self.deferred_wgrad is not None and not self.side_wgrad):
            for (spec, x, l), fn in zip(items, fns):
                self._conv_bias_bwd(spec, x, fn, rows_level=l)
            return
        asg, A, a0 = self.loss_rows
        cols = {k: [] for k in ("x", "ldx", "xc", "dy", "lddy", "a0", "w", "dx", "lddx", "acc", "slabs", "bacc", "h", "w_")}
        for (spec, x, l), fn in zip(items, fns):
            dyp, ld = fn()
            Ho, Wo = self.out_hw(spec, x)
            ns = self.L.dy_conv1x1_rows_slabs(x.N, Ho, Wo)
            slabs = self.transient((ns * spec.cout * spec.cin,), torch.float32)
            self.hold(slabs)
            self.deferred_wgrad.append((spec, slabs, ns, spec.acc_bias))
            acc = x.grad_target() if x.needs_grad else 0
            if x.needs_grad and not acc and self._sole_consumer_of_conv(x) is not None:
                self._rows_grad[(id(x.st), x.c0, x.C)] = (asg, A, a0[l])
            xp, xld, xc = self._src(x)
            for k, v in zip(cols, (xp, xld, xc, dyp, ld, a0[l], spec.weight.data_ptr(), x.gptr if x.needs_grad else 0, x.ld, acc,
                                   slabs.data_ptr(), self._acc_ready(spec.acc_bias), Ho, Wo)):
                cols[k].append(v)
        P, I = C.c_void_p, C.c_int
        spec0, x0 = items[0][0], items[0][1]
        self.call("dy_conv1x1_rows_backward_levels", len(items), self._arr(P, cols["x"]), self._arr(I, cols["ldx"]), self._arr(P, cols["xc"]),
                  self._arr(P, cols["dy"]), self._arr(I, cols["lddy"]), asg, A, self._arr(I, cols["a0"]), self._arr(P, cols["w"]),
                  self._arr(P, cols["dx"]), self._arr(I, cols["lddx"]), self._arr(I, cols["acc"]), self._arr(P, cols["slabs"]),
                  self._arr(P, cols["bacc"]), x0.N, self._arr(I, cols["h"]), self._arr(I, cols["w_"]), spec0.cin, spec0.cout)

    def head_cls_levels(self, items):
        """``conv_bias_cls`` for several levels at once (items = [(spec, x, y_ptr, dy_ptr_fn)], same cin / nc): one forward launch now,
        one backward launch first thing in the backward pass."""
        P, I, Lg = C.c_void_p, C.c_int, C.c_long
        srcs = [self._src(x) for _, x, _, _ in items]
        for _, x, _, _ in items:
            self._use(x)
        spec0 = items[0][0]
        self.call("dy_cls_head_forward_levels", len(items), self._arr(P, [s_[0] for s_ in srcs]), self._arr(I, [s_[1] for s_ in srcs]),
                  self._arr(P, [s_[2] for s_ in srcs]), self._arr(P, [sp.weight.data_ptr() for sp, _, _, _ in items]),
                  self._arr(P, [sp.bias.data_ptr() for sp, _, _, _ in items]), self._arr(P, [yp for _, _, yp, _ in items]),
                  self._arr(Lg, [x.npix for _, x, _, _ in items]), spec0.cin, spec0.cout)

        def bwd():
            assert self.deferred_wgrad is not None and not self.side_wgrad
            dys, dxs, accs, slabs_l, baccs = [], [], [], [], []
            ns = self.L.dy_cls_head_slabs()
            for spec, x, _, fn in items:
                dyp, ld = fn()
                assert ld == 8
                slabs = self.transient((ns * 16 * spec.cin,), torch.float32)
                self.hold(slabs)
                self.deferred_wgrad.append((spec, slabs, ns, spec.acc_bias))
                accs.append(x.grad_target() if x.needs_grad else 0)
                dys.append(dyp); dxs.append(x.gptr if x.needs_grad else 0); slabs_l.append(slabs.data_ptr())
                baccs.append(self._acc_ready(spec.acc_bias))
            self.call("dy_cls_head_backward_levels", len(items), self._arr(P, [s_[0] for s_ in srcs]), self._arr(I, [s_[1] for s_ in srcs]),
                      self._arr(P, [s_[2] for s_ in srcs]), self._arr(P, dys), self._arr(P, [sp.weight.data_ptr() for sp, _, _, _ in items]),
                      self._arr(P, dxs), self._arr(I, [x.ld for _, x, _, _ in items]), self._arr(I, accs), self._arr(P, slabs_l),
                      self._arr(P, baccs), self._arr(Lg, [x.npix for _, x, _, _ in items]), spec0.cin, spec0.cout)
        self.tape.append(bwd)

    def unapplied(self, x):
        """(raw Act, producing spec) when ``x`` was produced with ``defer_apply`` -- its own buffer holds nothing -- else None."""
        return self._unapplied.get((id(x.st), x.c0, x.C))

    def conv_bias_decode(self, spec: ConvSpec, x: Act, dy_ptr_fn, rows_level):
        """Detect's final box conv of level ``rows_level`` inside a StepPlan trace (caller checked ``rows_capable``): no logits are
        written; the plan launches dy_head_box_decode once the loss workspace is bound (``pending_decode``), the backward is the rows
        form."""
        self._use(x)
        self.rows_used.add(rows_level)
        self.pending_decode.append((spec, x, rows_level))
        self.tape.append(lambda: self._conv_bias_bwd(spec, x, dy_ptr_fn, rows_level=rows_level))

    infer_head = False  # set by InferPlan around an eval forward: Detect leaves its final convs + decode to dy_head_infer_levels
    pending_decode = None
    rows_used = None   # set() while a StepPlan traces its forward: the detection levels whose box conv can take the rows form
    loss_rows = None   # (assignment pointer, anchors per image, first anchor of each level) once the plan has bound the loss

    def _conv_bias_bwd(self, spec, x, dy_ptr_fn, accumulate=0, defer=True, rows_level=None):
        dyp, ld = dy_ptr_fn()
        Ho, Wo = self.out_hw(spec, x)
        if (rows_level is not None and self.loss_rows is not None and defer and not accumulate and self.deferred_wgrad is not None
                and not self.side_wgrad):
            asg, A, a0 = self.loss_rows
            ns = self.L.dy_conv1x1_rows_slabs(x.N, Ho, Wo)
            slabs = self.transient((ns * spec.cout * spec.cin,), torch.float32)
            self.hold(slabs)
            self.deferred_wgrad.append((spec, slabs, ns, spec.acc_bias))
            acc = x.grad_target() if x.needs_grad else 0
            if x.needs_grad and not acc and self._sole_consumer_of_conv(x) is not None:
                # x is a Conv's output whose only forward use was this conv: its gradient has the same rows, and nothing else
                self._rows_grad[(id(x.st), x.c0, x.C)] = (asg, A, a0[rows_level])
            src = self.unapplied(x)
            xp, xld, xcoef = (x.ptr, x.ld, 0) if src is None else (src[0].ptr, src[0].ld, src[1].coef.data_ptr())
            self.call("dy_conv1x1_rows_backward", xp, xld, xcoef, dyp, ld, asg, A, a0[rows_level], spec.weight.data_ptr(),
                      x.gptr if x.needs_grad else 0, x.ld, acc, slabs.data_ptr(), self._acc_ready(spec.acc_bias), x.N, Ho, Wo,
                      spec.cin, spec.cout)
            return
        assert self.unapplied(x) is None, f"{spec.name}: the dense backward needs the activated input that was never written"
        if (BIAS_WGRAD and defer and not accumulate and self.deferred_wgrad is not None and spec.acc_bias is not None and spec.ld is None
                and not self.side_wgrad):
            # the bias gradient rides on the weight-gradient kernel (sum of dY while it is staged) and the batched slab reduction
            self._conv_bwd(spec, x, dyp, ld, Ho, Wo, bias_acc=spec.acc_bias)
            return
        npix = x.N * Ho * Wo
        cp = spec.cout_phys
        part = self.scratch("partials", 2048 * 2 * cp * 4 + 4096)
        n = C.c_int(0)
        ident = self.ident_coef(cp)
        self.call("dy_bn_act_bwd_reduce", dyp, ld, dyp, ld, ident.data_ptr(), part.data_ptr(), 2048, npix, cp, DY_ACT_NONE,
                  C.byref(n))
        tmp = self.scratch("bwdcoef_tmp", 2 * cp * 4)
        self.call("dy_bn_bwd_finalize", part.data_ptr(), n.value, 0, spec.gbias.data_ptr(), tmp.data_ptr(), cp, 1.0, accumulate)
        self._conv_bwd(spec, x, dyp, ld, Ho, Wo, accumulate, defer=defer)

    def upsample2x(self, x: Act, out: Act | None = None):
        x = self.dense(x)
        if UPSEG and PLANAR and out is None and x.st.buf.dtype == torch.float16 and (self.tape is None or UPSEG_TRAIN):
            # left to the consumer: a Concat member read through the segment table (or Engine.dense, which runs the launch after all)
            up = UpAct(x)
            if self.tape is not None and x.needs_grad:
                self._use(x)

                def bwd():
                    if up._g is None or not up._g.st.gwritten:  # no segmented consumer wrote a gradient: it was executed after all
                        assert up._full is not None, "an up-sampled tensor without a consumer"  # (that launch's closure carries it)
                        return
                    assert up.grad_ready(), "gradient of an up-sampled concat member incomplete"
                    acc = x.grad_target()
                    self.call("dy_upsample2x", up.gptr, up.gld, x.gptr, x.ld, x.N, x.H, x.W, x.C, 1, acc)
                self.tape.append(bwd)
            return up
        return self._upsample_now(x, out)

    def _upsample_now(self, x: Act, out: Act | None = None, optional=False):
        """``optional``: the copy of an UpAct that something asked for after all (a per-layer capture, a consumer that is not a
        segmented conv) -- it may end up without a consumer of its own, i.e. without a gradient."""
        y = out if out is not None else self.new_act(x.N, 2 * x.H, 2 * x.W, x.C)
        self._use(x)
        self.call("dy_upsample2x", x.ptr, x.ld, y.ptr, y.ld, x.N, x.H, x.W, x.C, 0, 0)
        if self.tape is not None:
            def bwd():
                if optional and not y.st.gwritten:
                    return
                acc = x.grad_target()
                self.call("dy_upsample2x", y.gptr, y.ld, x.gptr, x.ld, x.N, x.H, x.W, x.C, 1, acc)
            self.tape.append(bwd)
        return y

    def maxpool5(self, x: Act, out: Act):
        x = self.dense(x)
        arg = self.transient((x.npix * x.C,), torch.uint8)
        self.hold(arg)
        self._use(x)
        self.call("dy_maxpool5", x.ptr, x.ld, out.ptr, out.ld, arg.data_ptr(), x.N, x.H, x.W, x.C)
        if self.tape is not None:
            def bwd():
                acc = x.grad_target()
                self.call("dy_maxpool5_backward", out.gptr, out.ld, arg.data_ptr(), x.gptr, x.ld, x.N, x.H, x.W, x.C, acc)
            self.tape.append(bwd)
        return out

    def sppf_pools(self, cat, c_):
        """SPPF's y1 = m(x), y2 = m(y1), y3 = m(y2) into slices 1..3 of ``cat`` (slice 0 = x): one launch with the map resident in LDS
        when it fits, three dy_maxpool5 launches otherwise.  Reference nn/modules/block.py:166-171."""
        x = cat.act(0, c_)
        if not (SPPF_FUSED and cat.buf.dtype == torch.float16 and self.L.dy_sppf_pool3_supported(x.H, x.W, c_)):
            for j in range(3):
                self.maxpool5(cat.act(j * c_, c_), cat.act((j + 1) * c_, c_))
            return
        sl = [cat.act(j * c_, c_) for j in range(4)]
        taping = self.tape is not None
        args = [self.transient((x.npix * c_,), torch.uint8) for _ in range(3)] if taping else [None] * 3
        if taping:
            self.hold(*args)
        for a in sl[:3]:
            self._use(a)
        ap = [t.data_ptr() if t is not None else 0 for t in args]
        self.call("dy_sppf_pool3", cat.act().ptr, cat.C, c_, ap[0], ap[1], ap[2], x.N, x.H, x.W)
        if taping:
            def bwd():
                acc = [0, 0, 0]
                for j in (2, 1, 0):  # the order the three stand-alone backward launches would claim their targets in
                    acc[j] = sl[j].grad_target()
                # (the closure holds the arg-max tensors themselves: un-recorded calls keep nothing alive otherwise)
                self.call("dy_sppf_pool3_backward", cat.act().gptr, cat.C, c_, args[0].data_ptr(), args[1].data_ptr(), args[2].data_ptr(),
                          x.N, x.H, x.W, acc[0], acc[1], acc[2])
            self.tape.append(bwd)

    def add(self, xs, out: Act | None = None):
        xs = [self.dense(t) for t in xs]
        a = xs[0]
        y = out if out is not None else self.new_act(a.N, a.H, a.W, a.C)
        b = xs[1]
        c = xs[2] if len(xs) > 2 else None
        assert len(xs) <= 3
        self._use(*xs)
        self.call("dy_add", a.ptr, a.ld, b.ptr, b.ld, 0 if c is None else c.ptr, 0 if c is None else c.ld, y.ptr, y.ld, a.npix, a.C)
        if self.tape is not None:
            def bwd():
                shared = False
                for t in xs:
                    if not t.needs_grad:
                        continue
                    if (ADD_ALIAS and not shared and t.st.gbuf is None and not t.st.gwritten and t.c0 == 0 and t.C == t.st.C
                            and y.c0 == 0 and y.C == y.st.C and y.st.gbuf is not None and tuple(t.st.buf.shape) == tuple(y.st.buf.shape)
                            and self._uses.get(id(t.st)) == {(t.c0, t.C): 1}):
                        # d(sum)/d(operand) is the identity: an operand whose ONLY forward use was this sum, and which has no gradient
                        # yet, takes the sum's gradient buffer as its own instead of a copy of it (nothing writes y's gradient after
                        # this point, and nothing else will write t's).  One operand per sum: two sharers could see each other's writers.
                        t.st.gbuf = y.st.gbuf
                        t.grad_target()
                        shared = True
                        continue
                    acc = t.grad_target()
                    if acc:
                        self.call("dy_add", t.gptr, t.ld, y.gptr, y.ld, 0, 0, t.gptr, t.ld, t.npix, t.C)
                    else:
                        self.call("dy_copy_slice", y.gptr, y.ld, t.gptr, t.ld, t.npix, t.C)
            self.tape.append(bwd)
        return y

    def concat(self, xs, out_storage=None):
        """Concat (reference nn/modules/conv.py:338-348).  Inputs already living in consecutive slices of one Storage
        are returned as a view; anything else is copied."""
        if any(isinstance(t, SegAct) for t in xs):
            xs = SegAct(xs).parts
        st = xs[0].st
        pos = xs[0].c0
        inplace = True
        for t in xs:
            if t.st is not st or t.c0 != pos:
                inplace = False
                break
            pos += t.C
        if inplace:
            return Act(st, xs[0].c0, pos - xs[0].c0)
        if PLANAR and len(xs) <= 8:
            return SegAct(xs)  # never materialised: the 1x1 conv behind it reads the members where they are
        return self._concat_copy(xs)

    def _concat_copy(self, xs):
        a = xs[0]
        y = self.new_act(a.N, a.H, a.W, sum(t.C for t in xs))
        self._use(*xs)
        off = 0
        parts = []
        for t in xs:
            if isinstance(t, UpAct):  # (inference) the up-sampling that was left to its consumer, written straight into its slice
                self.call("dy_upsample2x", t.src.ptr, t.src.ld, y.ptr + 2 * off, y.ld, t.src.N, t.src.H, t.src.W, t.C, 0, 0)
            else:
                self.call("dy_copy_slice", t.ptr, t.ld, y.ptr + 2 * off, y.ld, t.npix, t.C)
            parts.append((t, off))
            off += t.C
        if self.tape is not None:
            def bwd():
                for t, o in parts:
                    if not t.needs_grad:
                        continue
                    acc = t.grad_target()
                    gsrc = y.gptr + 2 * o
                    gl = t.gld if isinstance(t, UpAct) else t.ld  # (an UpAct's gradient tensor is full-resolution, of its own pitch)
                    if acc:
                        self.call("dy_add", t.gptr, gl, gsrc, y.ld, 0, 0, t.gptr, gl, t.npix, t.C)
                    else:
                        self.call("dy_copy_slice", gsrc, y.ld, t.gptr, gl, t.npix, t.C)
            self.tape.append(bwd)
        return y

    def zoom_cat(self, xs, out: Act | None = None):
        """Zoom_cat (reference nn/extra_modules/block.py:3402-3412): [maxpool+avgpool of the fine map | middle map | 2x
        nearest of the coarse map] written into one buffer (exact 2x pyramids only)."""
        l, m, s = (self.dense(t) for t in xs)
        assert l.H == 2 * m.H and l.W == 2 * m.W and m.H == 2 * s.H and m.W == 2 * s.W, "Zoom_cat needs exact 2x pyramids"
        y = out if out is not None else self.new_act(m.N, m.H, m.W, l.C + m.C + s.C)
        self._use(l, m, s)
        yl, ym, ys = y.sub(0, l.C), y.sub(l.C, m.C), y.sub(l.C + m.C, s.C)
        self.call("dy_zoom_pool", l.ptr, l.ld, yl.ptr, yl.ld, l.N, m.H, m.W, l.C)
        self.call("dy_copy_slice", m.ptr, m.ld, ym.ptr, ym.ld, m.npix, m.C)
        self.call("dy_upsample2x", s.ptr, s.ld, ys.ptr, ys.ld, s.N, s.H, s.W, s.C, 0, 0)
        if self.tape is not None:
            def bwd():
                if l.needs_grad:
                    acc = l.grad_target()
                    self.call("dy_zoom_pool_backward", l.ptr, l.ld, yl.gptr, yl.ld, l.gptr, l.ld, l.N, m.H, m.W, l.C, acc)
                if m.needs_grad:
                    acc = m.grad_target()
                    if acc:
                        self.call("dy_add", m.gptr, m.ld, ym.gptr, ym.ld, 0, 0, m.gptr, m.ld, m.npix, m.C)
                    else:
                        self.call("dy_copy_slice", ym.gptr, ym.ld, m.gptr, m.ld, m.npix, m.C)
                if s.needs_grad:
                    acc = s.grad_target()
                    self.call("dy_upsample2x", ys.gptr, ys.ld, s.gptr, s.ld, s.N, s.H, s.W, s.C, 1, acc)
            self.tape.append(bwd)
        return y

    # ---- ScalSeq (reference nn/extra_modules/block.py:3426-3443) --------------------------------------------------
    def scalseq(self, conv3d: ConvSpec, bn3d, coef, bwdcoef, gbn, ps, out: Act | None = None, res: Act | None = None):
        """ps = [p3 (full res), p4 (1/2), p5 (1/4)] already channel-matched.  conv3d: 1x1x1 conv with bias applied to
        every scale at its native resolution (a 1x1 conv commutes with nearest up-sampling); BatchNorm3d statistics
        are those of the up-sampled (B,3,H,W) volume, i.e. level l weighs 4**l.  ``res``: a tensor added to the result (the ``Add`` that
        follows ScalSeq in the ASF models, reference nn/extra_modules/block.py:3479-3484, folded into the tail kernel)."""
        ps = [self.dense(t) for t in ps]
        p3 = ps[0]
        N, H, W, Cc = p3.N, p3.H, p3.W, conv3d.cout
        assert ps[1].H * 2 == H and ps[2].H * 4 == H and ps[1].W * 2 == W and ps[2].W * 4 == W, "ScalSeq needs exact 2x/4x pyramids"
        raws, nps, offs = [], [], []
        cp16 = (Cc + 15) // 16 * 16
        total = 0
        for l, p in enumerate(ps):
            n = self.L.dy_conv_num_partials(p.N, p.H, p.W, p.C, Cc, 1, 1, 1)
            nps.append(n)
            offs.append(total)
            total += n * 2 * cp16 * 4
        part = self.scratch("partials_ss", total + 4096)
        for l, p in enumerate(ps):
            r = self.new_act(p.N, p.H, p.W, Cc)
            epi = DY_EPI_BIAS | (DY_EPI_STATS if self.training else 0)
            self._conv_raw(conv3d, p, r.ptr, r.ld, epi, part.data_ptr() + offs[l], conv3d.bias)
            raws.append(r)
        if self.training:
            self.call("dy_bn_finalize", part.data_ptr() + offs[0], nps[0], 1.0, part.data_ptr() + offs[1], nps[1], 4.0,
                      part.data_ptr() + offs[2], nps[2], 16.0, bn3d["weight"].data_ptr(), bn3d["bias"].data_ptr(),
                      bn3d["running_mean"].data_ptr(), bn3d["running_var"].data_ptr(), coef.data_ptr(), Cc,
                      float(3 * N * H * W), BN3D_EPS, BN3D_MOM, 1)
        else:
            self.call("dy_bn_eval_coef", bn3d["weight"].data_ptr(), bn3d["bias"].data_ptr(), bn3d["running_mean"].data_ptr(),
                      bn3d["running_var"].data_ptr(), coef.data_ptr(), Cc, BN3D_EPS)
        y = out if out is not None else self.new_act(N, H, W, Cc)
        if res is not None:
            assert (res.N, res.H, res.W, res.C) == (N, H, W, Cc)
            self._use(res)
        self.call("dy_scalseq_tail", raws[0].ptr, raws[0].ld, raws[1].ptr, raws[1].ld, raws[2].ptr, raws[2].ld,
                  0 if res is None else res.ptr, 0 if res is None else res.ld, y.ptr, y.ld, coef.data_ptr(), N, H, W, Cc)
        if self.tape is not None:
            def bwd():
                if res is not None and res.needs_grad:  # d(sum)/d(res) = identity, as in Engine.add
                    if (ADD_ALIAS and res.st.gbuf is None and not res.st.gwritten and res.c0 == 0 and res.C == res.st.C and y.c0 == 0
                            and y.C == y.st.C and y.st.gbuf is not None and tuple(res.st.buf.shape) == tuple(y.st.buf.shape)):
                        # res has no gradient yet: it takes y's gradient buffer as its own.  Every writer of y's gradient has run, its
                        # one reader (the ScalSeq backward below) runs before anything can be added to res's gradient.
                        res.st.gbuf = y.st.gbuf
                        res.grad_target()
                    elif res.grad_target():
                        self.call("dy_add", res.gptr, res.ld, y.gptr, y.ld, 0, 0, res.gptr, res.ld, res.npix, res.C)
                    else:
                        self.call("dy_copy_slice", y.gptr, y.ld, res.gptr, res.ld, res.npix, res.C)
                self._scalseq_bwd(conv3d, coef, bwdcoef, gbn, ps, raws, y)
            self.tape.append(bwd)
        return y

    def _scalseq_bwd(self, conv3d, coef, bwdcoef, gbn, ps, raws, y):
        assert y.grad_ready()
        N, H, W, Cc = y.N, y.H, y.W, y.C
        part = self.scratch("partials_ss", 2048 * 2 * Cc * 4 + 4096)
        rargs = (raws[0].ptr, raws[0].ld, raws[1].ptr, raws[1].ld, raws[2].ptr, raws[2].ld, y.gptr, y.ld)
        # one pass over dY / r0 / r1 / r2 per mode (all three levels at once)
        n = C.c_int(0)
        self.call("dy_scalseq_tail_backward_all", *rargs, 0, 0, 0, 0, 0, 0, coef.data_ptr(), 0, part.data_ptr(), 2048, N, H, W, Cc, 0,
                  C.byref(n))
        self.call("dy_bn_bwd_finalize", part.data_ptr(), n.value, gbn[0].data_ptr(), gbn[1].data_ptr(), bwdcoef.data_ptr(), Cc,
                  float(3 * N * H * W), 0)
        drs = [self.scratch(f"draw_ss{l}", ps[l].npix * Cc * 2) for l in range(3)]
        self.call("dy_scalseq_tail_backward_all", *rargs, drs[0].data_ptr(), Cc, drs[1].data_ptr(), Cc, drs[2].data_ptr(), Cc,
                  coef.data_ptr(), bwdcoef.data_ptr(), 0, 0, N, H, W, Cc, 1, None)
        for l in range(3):
            # shared conv3d weights/bias: the three scales accumulate into one gradient
            self._conv_bias_bwd(conv3d, ps[l], lambda dr=drs[l], Cc=Cc: (dr.data_ptr(), Cc), accumulate=1 if l else 0, defer=False)

    # ---- LDConv (reference nn/modules/conv.py:366-410) -----------------------------------------------------------
    def ldconv(self, sp_p: ConvSpec, sp_c: ConvSpec, pn_i32, Np, stride, x: Act, out: Act | None = None):
        x = self.dense(x)
        h, w = self.out_hw(sp_p, x)
        off = self.transient((x.N, h, w, 2 * Np), torch.float32)
        self.hold(off)
        doff = None
        if self.tape is not None:
            doff = torch.zeros((x.N, h, w, 8 * ((2 * Np + 7) // 8)), dtype=torch.float16, device=self.device)
            self.hold(doff)
        self.conv_bias(sp_p, x, off.data_ptr(), 2 * Np, True, lambda: (doff.data_ptr(), doff.shape[-1]))
        self._use(x)  # the sampler's backward writes x's gradient too
        xo = self.new_act(x.N, h, w, Np * x.C)
        self.call("dy_ldconv_sample", x.ptr, x.ld, off.data_ptr(), 2 * Np, pn_i32.data_ptr(), xo.ptr, xo.ld, x.N, x.H, x.W, h, w,
                  x.C, Np, stride)
        if self.tape is not None:
            def bwd():
                assert xo.grad_ready()
                if not x.needs_grad:  # stem: offset gradient only
                    self.call("dy_ldconv_sample_backward", x.ptr, x.ld, off.data_ptr(), 2 * Np, pn_i32.data_ptr(), xo.gptr, xo.ld,
                              0, doff.data_ptr(), doff.shape[-1], x.N, x.H, x.W, h, w, x.C, Np, stride)
                    return
                dx32 = self.scratch("ld_dx32", x.npix * x.C * 4)  # touched only when the offsets outgrow the gather radius
                acc = x.grad_target()
                if os.environ.get("DY_LD_SCATTER", "0") == "1":  # measurement switch: the atomic path on its own
                    self.call("dy_fill_zero", dx32.data_ptr(), x.npix * x.C * 4)
                    self.call("dy_ldconv_sample_backward", x.ptr, x.ld, off.data_ptr(), 2 * Np, pn_i32.data_ptr(), xo.gptr, xo.ld,
                              dx32.data_ptr(), doff.data_ptr(), doff.shape[-1], x.N, x.H, x.W, h, w, x.C, Np, stride)
                    self.call("dy_f32_to_f16_add", dx32.data_ptr(), x.gptr, x.ld, x.npix, x.C, acc)
                    return
                self.call("dy_ldconv_sample_backward_gather", x.ptr, x.ld, off.data_ptr(), 2 * Np, pn_i32.data_ptr(), xo.gptr, xo.ld,
                          x.gptr, x.ld, acc, dx32.data_ptr(), doff.data_ptr(), doff.shape[-1], self.scratch("ld_maxabs", 16).data_ptr(),
                          int(os.environ.get("DY_LD_RMAX", "2")), x.N, x.H, x.W, h, w, x.C, Np, stride)
            self.tape.append(bwd)
        return self.conv_bn_act(sp_c, xo, out)

    # ---- loss -----------------------------------------------------------------------------------------------------
    def zero_f32(self, t):
        self.call("dy_fill_zero", t.data_ptr(), t.numel() * t.element_size())
