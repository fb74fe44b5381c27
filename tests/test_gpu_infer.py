"""-m gpu: the eval forward as a recorded plan (ultralytics/hip/infer.py; reference get_FPS.py:42-87, engine/predictor.py:150) and
Detect's fused inference tail (csrc/head_infer.hip; reference nn/modules/head.py:50-74).

What is pinned: the fused tail produces the BITS of its unfused form (dy_conv_forward with fp32 output + bias, then
dy_decode_predictions) on the same inputs; a plan -- traced, replayed, captured into a hipGraph -- produces the bits of the same
launches issued by walking the modules; the result is the caller's own tensor; the per-level feature maps are still available
lazily; plans follow the model's geometry, weights and parameter storage."""
import ctypes as C
import os

import pytest
import torch

from conftest import CFG_DIR
from gpu_util import relerr

pytestmark = pytest.mark.gpu


def _model(name, seed=11, fuse=False):
    from oracle import graph as og
    from ultralytics.nn.tasks import DetectionModel
    p = os.path.join(CFG_DIR, name + ".yaml")
    m = DetectionModel(p, ch=3, verbose=False)
    g = og.build_graph(og.load_yaml(p))
    m.load_state_dict(og.fill_state(og.state_layout(g), seed), strict=True)
    m = m.cuda().eval()
    return m.fuse() if fuse else m


@pytest.mark.parametrize("cin_cls,nc", [(32, 6), (64, 20), (80, 80), (128, 1), (48, 17)])
def test_fused_tail_is_conv_plus_decode_bit_for_bit(cin_cls, nc):
    """dy_head_infer_levels against its unfused form on three levels (one of them ending mid-wave, B*H*W not a multiple of 64), with
    channel-sliced inputs (ld > C): y must be EQUAL -- same MFMA chunk order as dy_conv_forward's geometry for that channel count
    (16-channel half-empty steps for 48 / 80), same softmax / expectation association as decode_pred_kernel."""
    from ultralytics.hip import DY_EPI_BIAS, DY_EPI_F32OUT, check, lib
    from ultralytics.hip.engine import ConvSpec, Engine
    L, eng = lib(), Engine("cuda:0")
    assert L.dy_head_infer_supported(64, 64, cin_cls, nc) == 1
    gen = torch.Generator().manual_seed(cin_cls * 100 + nc)
    B, hw, strides = 3, [(12, 20), (6, 10), (3, 5)], [8.0, 16.0, 32.0]
    ncp = (nc + 7) // 8 * 8
    xs, specs, box, cls = [], [], [], []
    for h, w in hw:
        xb = (torch.randn(B, h, w, 64 + 16, generator=gen) * 1.5).half().cuda()      # the box input is a 64-channel slice of an 80-wide buffer
        xc = (torch.randn(B, h, w, cin_cls + 8, generator=gen) * 1.5).half().cuda()
        wb, bb = (torch.randn(64, 64, 1, 1, generator=gen) * 0.2).cuda(), torch.randn(64, generator=gen).cuda()
        wc, bc = (torch.randn(nc, cin_cls, 1, 1, generator=gen) * 0.2).cuda(), torch.randn(nc, generator=gen).cuda()
        sb, sc = ConvSpec("b", wb, bb, None, 1, 1, 0), ConvSpec("c", wc, bc, None, 1, 1, 0)
        for sp in (sb, sc):
            eng.prepare_conv(sp)
            eng.pack(sp, transposed=False)
        ab, ac = eng.wrap_act(xb).sub(8, 64), eng.wrap_act(xc).sub(0, cin_cls)
        fb = torch.empty(B, h, w, 64, dtype=torch.float32, device="cuda")
        fc = torch.zeros(B, h, w, ncp, dtype=torch.float32, device="cuda")
        eng._conv_raw(sb, ab, fb.data_ptr(), 64, DY_EPI_BIAS | DY_EPI_F32OUT, 0, bb)
        eng._conv_raw(sc, ac, fc.data_ptr(), ncp, DY_EPI_BIAS | DY_EPI_F32OUT, 0, bc)
        xs.append((ab, ac)); specs.append((sb, sc)); box.append(fb); cls.append(fc)
    nl, A = len(hw), sum(h * w for h, w in hw)
    s = torch.cuda.current_stream().cuda_stream
    P, I, F = C.c_void_p, C.c_int, C.c_float
    arr = lambda t, v: (t * len(v))(*v)  # noqa: E731
    want = torch.empty(B, 4 + nc, A, dtype=torch.float32, device="cuda")
    check(L.dy_decode_predictions(arr(P, [t.data_ptr() for t in box]), arr(P, [t.data_ptr() for t in cls]), arr(I, [h for h, _ in hw]),
                                  arr(I, [w for _, w in hw]), arr(F, strides), nl, B, nc, ncp, want.data_ptr(), s), "dy_decode_predictions")
    got = torch.full((B, 4 + nc, A), float("nan"), dtype=torch.float32, device="cuda")
    check(L.dy_head_infer_levels(nl, arr(P, [a.ptr for a, _ in xs]), arr(I, [a.ld for a, _ in xs]), arr(P, [sb.weight.data_ptr() for sb, _ in specs]),
                                 arr(P, [sb.bias.data_ptr() for sb, _ in specs]), arr(P, [c.ptr for _, c in xs]), arr(I, [c.ld for _, c in xs]),
                                 arr(P, [sc.weight.data_ptr() for _, sc in specs]), arr(P, [sc.bias.data_ptr() for _, sc in specs]),
                                 arr(I, [h for h, _ in hw]), arr(I, [w for _, w in hw]), arr(F, strides), B, cin_cls, nc, got.data_ptr(), s),
          "dy_head_infer_levels")
    torch.cuda.synchronize()
    assert torch.isfinite(got).all()
    assert torch.equal(got, want), f"max abs diff {float((got - want).abs().max()):.3e}"


@pytest.mark.parametrize("name,fuse,size", [("yolov8n-ASF-P2P2", False, 64), ("yolov8n-ASF-P2P2", True, 96), ("yolov8n-LD-P2", False, 64),
                                            ("yolov8n-p2", True, 128), ("yolov8n-ASF-P2", True, 64)])
def test_plan_replays_the_walked_forward_bit_for_bit(name, fuse, size):
    from ultralytics.hip import infer as I
    m = _model(name, fuse=fuse)
    x = torch.rand(2, 3, size, size, generator=torch.Generator().manual_seed(5)).cuda()
    with torch.no_grad():
        y1, f1 = m(x)   # first sight of the geometry: the modules are walked (stem from the image batch, fused tail)
        assert not m._infer_plans["plans"]
        y2, f2 = m(x)   # second: traced into a plan (and captured when the runtime allows graphs)
        plan = m._infer_plans["plans"][(2, 3, size, size)]
        y3, f3 = m(x)   # replay
        y4, _ = m(x * 0.5)
        y5, _ = m(x)
    torch.cuda.synchronize()
    assert plan.calls == 4 and (plan.graph is not None) == plan.use_graph
    # the stem is launched on the caller's tensor: the forward of x * 0.5 never touched the plan's static input ...
    assert plan.head >= 1 and torch.equal(plan.img, x)
    with torch.no_grad():  # ... while an input the stem cannot read as it is (fp16, or a strided view) goes through it
        y6, _ = m(x.half())
        assert torch.equal(plan.img, x.half().float())
        y7, _ = m(torch.stack([x, x], -1)[..., 0])
        assert torch.equal(plan.img, x)
    assert torch.equal(y7, y1) and relerr(y6, y1) < 2e-2
    assert torch.equal(y1, y2) and torch.equal(y1, y3) and torch.equal(y1, y5) and not torch.equal(y1, y4)
    assert len({t.data_ptr() for t in (y1, y2, y3, y4, y5)}) == 5, "every forward returns a tensor of its own"
    # the per-level (B, no, H, W) maps of the reference's inference return (head.py:74), materialised on demand
    feats = list(f3)
    nc = m.model[-1].nc
    assert len(feats) == len(m.model[-1].stride) and all(f.shape[1] == nc + 64 for f in feats)
    # ... and against the path this round replaced (import pass, generic final convs, dy_decode_predictions): the stem sums its 27
    # taps in another order, everything else is the same arithmetic
    I.INFER_PLAN = False
    try:
        with torch.no_grad():
            y0, f0 = m(x)
    finally:
        I.INFER_PLAN = True
    e = relerr(y1, y0)
    ef = max(relerr(a.float(), b.float()) for a, b in zip(feats, list(f0)))
    print(f"{name} fuse={fuse}: y vs the un-planned path {e:.2e}, feature maps {ef:.2e}")
    assert e < (2e-2 if "LD" in name else 2e-3) and ef < (5e-2 if "LD" in name else 5e-3)


def test_plans_follow_geometry_weights_and_parameter_storage():
    from ultralytics.hip import infer as I
    m = _model("yolov8n-ASF-P2P2")
    xa, xb = torch.rand(1, 3, 64, 64).cuda(), torch.rand(2, 3, 96, 64).cuda()
    with torch.no_grad():
        for _ in range(3):
            ya, yb = m(xa)[0], m(xb)[0]
        assert set(m._infer_plans["plans"]) == {(1, 3, 64, 64), (2, 3, 96, 64)}
        assert ya.shape[0] == 1 and yb.shape == (2, 10, 24 * 16 + 12 * 8 + 6 * 4)
        # a weight edited in place is picked up by the next replay (the plan re-packs, as the walked forward does)
        w = m.model[-1].cv3[0][2].bias
        w0 = w.data.clone()
        w.data.add_(1.0)
        yc = m(xa)[0]
        assert not torch.equal(yc[:, 4:], ya[:, 4:]) and torch.equal(yc[:, :4], ya[:, :4])
        w.data.copy_(w0)
        assert torch.equal(m(xa)[0], ya)
        # fuse() re-creates parameters: every recorded pointer is stale, the plans are dropped and rebuilt on the new runtime
        m.fuse()
        y1 = m(xa)[0]
        assert not m._infer_plans["plans"]
        y2 = m(xa)[0]
        assert set(m._infer_plans["plans"]) == {(1, 3, 64, 64)} and torch.equal(y1, y2)
        assert relerr(y1, ya) < 5e-3  # BatchNorm folded into fp16 weights instead of applied in fp32
        # more geometries than DY_INFER_PLANS: the least recently used plan goes
        for s in (32, 64, 96, 128, 160):
            xs = torch.rand(1, 3, s, s).cuda()
            m(xs), m(xs)
        assert len(m._infer_plans["plans"]) == I.MAX_PLANS and (1, 3, 160, 160) in m._infer_plans["plans"]
    # training mode never takes the plan
    m.train()
    out = m(xa)
    assert isinstance(out, list) and out[0].shape[1] == 70


def test_split_tail_cout_group_is_the_same_convolution(monkeypatch):
    """Fused 3x3 convs with 64 m + 16 outputs (Detect's class branch at nc = 80) run their last 16 channels as a launch of their own
    (nn/modules/conv.py, DY_SPLIT_COUT) instead of a zero-padded 64-wide group: every output channel is the same dot product, summed
    in the chunk order the 16-row geometry takes (64-channel chunks where the 64-row group takes 32-channel ones): fp32 rounding
    order: 1e-8 of the largest output measured, 1e-5 the bound."""
    from ultralytics.hip import infer as I
    from ultralytics.nn.modules import conv as conv_mod
    x = torch.rand(2, 3, 128, 128, generator=torch.Generator().manual_seed(9)).cuda()
    outs = []
    for on in (True, False):
        monkeypatch.setattr(conv_mod, "SPLIT_COUT", on)
        m = _model("yolov8n-p2", fuse=True)
        with torch.no_grad():
            outs.append(m(x)[0])
        n_split = sum(1 for mod in m.modules() if mod.__dict__.get("_split") is not None)
        assert (n_split == 8) == on, n_split  # cv3[l][0] and cv3[l][1] of the four levels
    e = relerr(outs[0], outs[1])
    print(f"split vs padded tail group: {e:.2e}")
    assert e < 1e-5


@pytest.mark.parametrize("N,h,w,ca,cb,cout", [(9, 3, 3, 32, 32, 32), (2, 20, 20, 64, 64, 64), (3, 5, 7, 32, 64, 48), (1, 40, 40, 128, 64, 128),
                                              (5, 2, 8, 64, 128, 64)])
def test_upsampled_concat_member_is_read_where_it_lies(N, h, w, ca, cb, cout):
    """nn.Upsample(None, 2, 'nearest') in front of a Concat (reference cfg/models/*.yaml, top-down path) left to the 1x1 conv behind
    the concatenation (engine.UpAct, DySegs.acc = 2): the staging reads pixel (y >> 1, x >> 1) of the low-resolution tensor -- the
    same values the materialised copy holds, so the output must be EQUAL; maps smaller than a 256-pixel tile (a tile then spans
    several images) and channel-sliced members included."""
    from ultralytics.hip.engine import ConvSpec, Engine, SegAct, Storage, UpAct
    eng = Engine("cuda:0")
    eng.training = False
    g = torch.Generator().manual_seed(N * 100 + h * 10 + w)
    wt = (torch.randn(cout, ca + cb, 1, 1, generator=g) / (ca + cb) ** 0.5).cuda()
    sp = ConvSpec("c", wt, torch.randn(cout, generator=g).cuda(), None, 1, 1, 1)
    eng.prepare_conv(sp)
    eng.pack(sp, transposed=False)
    a = Storage(eng, N, h, w, ca + 16)     # the low-resolution member is a channel slice of a wider buffer
    a.buf.copy_(torch.randn(N, h, w, ca + 16, generator=g).half())
    b = Storage(eng, N, 2 * h, 2 * w, cb)
    b.buf.copy_(torch.randn(N, 2 * h, 2 * w, cb, generator=g).half())
    aa = a.act(8, ca)
    seg = SegAct([UpAct(aa), b.act()])
    if not eng.seg_conv_ok(sp, seg):
        pytest.skip("this shape is not taken by the segmented kernel")
    y1 = eng.conv_fused(sp, seg)
    y0 = eng.conv_fused(sp, eng._concat_copy([eng._upsample_now(aa), b.act()]))
    torch.cuda.synchronize()
    assert torch.isfinite(y1.st.buf.float()).all() and float(y1.st.buf.float().abs().max()) > 0
    assert torch.equal(y1.st.buf, y0.st.buf), f"max diff {float((y1.st.buf.float() - y0.st.buf.float()).abs().max()):.3e}"


@pytest.mark.parametrize("name,fuse,size", [("yolov8n-p2", True, 128), ("yolov8n-ASF-P2P2", False, 96), ("yolov8n-LD-P2", True, 64)])
def test_models_without_the_upsample_launch_give_the_same_bits(name, fuse, size, monkeypatch):
    from ultralytics.hip import engine as E
    x = torch.rand(2, 3, size, size, generator=torch.Generator().manual_seed(3)).cuda()
    outs = []
    for on in (False, True):
        monkeypatch.setattr(E, "UPSEG", on)
        m = _model(name, fuse=fuse)
        with torch.no_grad():
            m(x)
            y, _ = m(x)
        plan = m._infer_plans["plans"][(2, 3, size, size)]
        n_up = sum(1 for o in plan.rec.ops if o[2] == "dy_upsample2x")
        assert (n_up == 0) == on, n_up
        outs.append(y)
    assert torch.equal(outs[0], outs[1])
