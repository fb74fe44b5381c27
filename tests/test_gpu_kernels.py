"""-m gpu: each HIP kernel family against a plain PyTorch fp32 reference of the same op (inputs pre-rounded to fp16, the
storage format of the path; accumulation is fp32 on both sides).  Tolerances are relative to the largest reference
magnitude: 2e-3 covers the single fp16 rounding of the output, 5e-3 two roundings (bn/act chains)."""
import ctypes as C
import os

import pytest
import torch
import torch.nn.functional as F

from gpu_util import h16, relerr

pytestmark = pytest.mark.gpu


def _eng():
    from ultralytics.hip.engine import Engine
    return Engine("cuda:0")


def _spec(eng, w, bias, ks, s):
    from ultralytics.hip.engine import ConvSpec
    sp = ConvSpec("t", w.cuda().contiguous(), None if bias is None else bias.cuda(), None, ks, s, 0)
    sp.gweight = torch.zeros_like(sp.weight)
    sp.gbias = torch.zeros(((sp.cout + 7) // 8 * 8), device="cuda") if bias is not None else None
    eng.prepare_conv(sp)
    eng.pack(sp, transposed=True)
    return sp


def _act(eng, x):
    from ultralytics.hip.engine import Storage
    N, Cc, H, W = x.shape
    cp = (Cc + 7) // 8 * 8
    st = Storage(eng, N, H, W, cp)
    st.buf.zero_()
    st.buf[..., :Cc].copy_(x.cuda().permute(0, 2, 3, 1))
    return st.act()


CONV_CASES = [  # cin, cout, ks, s, H, W
    (16, 32, 3, 1, 12, 20), (64, 64, 3, 1, 40, 40), (32, 64, 3, 2, 26, 34), (64, 128, 3, 2, 16, 16), (128, 64, 3, 1, 9, 11),
    (3, 16, 3, 2, 32, 48), (16, 32, 3, 2, 13, 21), (48, 32, 1, 1, 7, 9), (256, 128, 1, 1, 10, 10), (96, 64, 1, 1, 13, 5), (32, 6, 1, 1, 8, 8),
    (16, 16, 3, 1, 33, 35), (64, 80, 1, 1, 6, 6), (192, 128, 1, 1, 5, 7),
    # padded cout group (48 -> MT = 4), 128-channel layers with narrowed cout groups, parity-class stride-2 dgrad at odd sizes
    (32, 48, 3, 1, 10, 14), (128, 128, 3, 1, 12, 12), (64, 64, 3, 2, 15, 17), (128, 64, 3, 2, 9, 9), (64, 48, 1, 1, 11, 3),
    # maps 40 / 80 pixels wide (full-width tiles), heights that end inside a wave's strip
    (32, 32, 3, 1, 37, 40), (128, 64, 3, 1, 11, 40), (64, 32, 3, 1, 5, 80),
]


@pytest.mark.parametrize("cin,cout,ks,s,H,W", CONV_CASES)
def test_conv_forward_dgrad_wgrad(cin, cout, ks, s, H, W):
    torch.manual_seed(cin * 1000 + cout + ks + s)
    eng = _eng()
    N = 2
    x = h16(torch.randn(N, cin, H, W))
    w = h16(torch.randn(cout, cin, ks, ks) / (cin * ks * ks) ** 0.5)
    b = torch.randn(cout)
    sp = _spec(eng, w, b, ks, s)
    xa = _act(eng, x)
    Ho, Wo = eng.out_hw(sp, xa)
    ref = F.conv2d(x, w, b, s, ks // 2)
    # fp32 output path with bias (Detect heads)
    y32 = torch.zeros(N, Ho, Wo, cout, device="cuda")
    eng.conv_bias(sp, xa, y32.data_ptr(), cout, True)
    torch.cuda.synchronize()
    assert relerr(y32.permute(0, 3, 1, 2), ref) < 2e-5 * (cin * ks * ks) ** 0.5 + 1e-5
    # fp16 output, no bias, with batch statistics partials
    cp = (cout + 15) // 16 * 16
    if cout % 8 == 0:
        y16 = torch.zeros(N, Ho, Wo, cout, dtype=torch.float16, device="cuda")
        nparts = eng.L.dy_conv_num_partials(N, H, W, xa.C, cout, ks, s, 1)
        part = torch.zeros(nparts, 2, cp, device="cuda")
        eng.call("dy_conv_forward", xa.ptr, xa.ld, sp.wpack.data_ptr(), 0, y16.data_ptr(), cout, part.data_ptr(), N, H, W, xa.C, cout,
                 ks, s, 1, 0, 0, 1, None)
        torch.cuda.synchronize()
        ref0 = F.conv2d(x, w, None, s, ks // 2)
        assert relerr(y16.float().permute(0, 3, 1, 2), ref0) < 2e-3
        yq = y16.float()
        # BN partial sums: the ping-pong kernel sums its fp32 accumulators (== the fp32 conv up to summation order), the
        # older kernels sum the fp16-rounded stored values; both must sit within fp16 rounding noise of the fp32 sums
        r0 = ref0.permute(0, 2, 3, 1)
        assert relerr(part[:, 0, :cout].sum(0), r0.sum((0, 1, 2))) < 5e-4
        assert relerr(part[:, 1, :cout].sum(0), (r0 * r0).sum((0, 1, 2))) < 5e-4
        assert min(relerr(part[:, 0, :cout].sum(0), r0.sum((0, 1, 2))), relerr(part[:, 0, :cout].sum(0), yq.sum((0, 1, 2)))) < 1.1e-4
    # gradients: dY random (fp16), dX via transposed pack (+ zero-dilation for stride 2), dW via the tr-read kernel
    coutp = (cout + 7) // 8 * 8
    dy = h16(torch.randn(N, cout, Ho, Wo))
    dya = _act(eng, dy)
    xa.needs_grad = cin != 3
    xr = x.clone().requires_grad_(True)
    wr = w.clone().requires_grad_(True)
    F.conv2d(xr, wr, None, s, ks // 2).backward(dy)
    eng._conv_bwd(sp, xa, dya.ptr, dya.ld, Ho, Wo)
    torch.cuda.synchronize()
    assert relerr(sp.gweight, wr.grad) < 3e-4, "wgrad"
    if xa.needs_grad:
        gx = xa.st.gbuf[..., :cin].permute(0, 3, 1, 2).float()
        assert relerr(gx, xr.grad) < 2e-3, "dgrad"
        # accumulate path: a second writer adds on top
        eng._conv_bwd(sp, xa, dya.ptr, dya.ld, Ho, Wo, accumulate_w=1)
        torch.cuda.synchronize()
        assert relerr(xa.st.gbuf[..., :cin].permute(0, 3, 1, 2).float(), 2 * xr.grad) < 3e-3
        assert relerr(sp.gweight, 2 * wr.grad) < 3e-4
    _ = coutp


def test_full_width_tiles_give_the_bits_of_the_32_column_tiles(tmp_path):
    """conv_mfma_pp_kernel's full-width tiles (maps exactly 40 / 80 pixels wide: a wave owns an 80-pixel strip instead of 2 x 32
    columns, csrc/conv.hip FW) change WHICH pixels a wave computes, not how a pixel is computed: chunk / tap order per output is the
    same, so every epilogue's output must be EQUAL to the 32-column scheme's -- plain store, BatchNorm statistics (the sums to fp32
    rounding: a workgroup's fp32 partial sums cover other pixels, 6e-8 measured), bias + SiLU, accumulate, residual, fp32 + bias;
    heights that end inside a strip included."""
    import subprocess
    import sys
    outs = {}
    for fw in ("0", "80"):
        f = tmp_path / f"fw{fw}.pt"
        env = dict(os.environ, DY_CONV_FW=fw)
        r = subprocess.run([sys.executable, os.path.join(os.path.dirname(__file__), "fw_worker.py"), str(f)], env=env, capture_output=True,
                           text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        outs[fw] = torch.load(f)
    assert outs["0"].keys() == outs["80"].keys() and len(outs["0"]) > 50
    for k, a in outs["0"].items():
        b = outs["80"][k]
        assert torch.isfinite(a.float()).all() and float(a.float().abs().max()) > 0, k
        if k.endswith("stats_acc"):
            assert relerr(b, a) < 1e-6, k
        else:
            assert torch.equal(a, b), f"{k}: max diff {float((a.float() - b.float()).abs().max()):.3e}"


@pytest.mark.parametrize("C_,act", [(16, 1), (64, 1), (32, 2), (128, 0)])
def test_bn_act_forward_backward(C_, act):
    torch.manual_seed(C_ + act)
    eng = _eng()
    N, H, W = 3, 9, 14
    x = h16(torch.randn(N, C_, H, W) * 1.5 + 0.3)
    gamma, beta = torch.rand(C_) + 0.5, torch.randn(C_) * 0.2
    rm, rv = torch.zeros(C_), torch.ones(C_)
    dy = h16(torch.randn(N, C_, H, W))
    xr = x.clone().requires_grad_(True)
    gr, br = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    rm_ref, rv_ref = rm.clone(), rv.clone()
    z = F.batch_norm(xr, rm_ref, rv_ref, gr, br, True, 0.03, 1e-3)
    yr = F.silu(z) if act == 1 else (F.leaky_relu(z, 0.1) if act == 2 else z)
    yr.backward(dy)
    xa, dya = _act(eng, x), _act(eng, dy)
    npix = N * H * W
    # statistics from per-"workgroup" partials: emulate with one partial row
    xq = xa.st.buf.float().view(-1, C_)
    part = torch.stack([xq.sum(0), (xq * xq).sum(0)]).view(1, 2, C_).contiguous()
    g_, b_, rm_, rv_ = gamma.cuda(), beta.cuda(), rm.cuda(), rv.cuda()
    coef, bwd = torch.zeros(4 * C_, device="cuda"), torch.zeros(2 * C_, device="cuda")
    eng.call("dy_bn_finalize", part.data_ptr(), 1, 1.0, 0, 0, 0.0, 0, 0, 0.0, g_.data_ptr(), b_.data_ptr(), rm_.data_ptr(),
             rv_.data_ptr(), coef.data_ptr(), C_, float(npix), 1e-3, 0.03, 1)
    y = torch.zeros(N, H, W, C_, dtype=torch.float16, device="cuda")
    eng.call("dy_bn_act_apply", xa.ptr, xa.ld, 0, 0, y.data_ptr(), C_, coef.data_ptr(), npix, C_, act)
    torch.cuda.synchronize()
    assert relerr(y.float().permute(0, 3, 1, 2), yr) < 2e-3
    assert relerr(rm_, rm_ref) < 1e-5 and relerr(rv_, rv_ref) < 1e-5
    parts = torch.zeros(2048 * 2 * C_, device="cuda")
    n = C.c_int(0)
    eng.call("dy_bn_act_bwd_reduce", dya.ptr, dya.ld, xa.ptr, xa.ld, coef.data_ptr(), parts.data_ptr(), 2048, npix, C_, act,
             C.byref(n))
    dg, db = torch.zeros(C_, device="cuda"), torch.zeros(C_, device="cuda")
    eng.call("dy_bn_bwd_finalize", parts.data_ptr(), n.value, dg.data_ptr(), db.data_ptr(), bwd.data_ptr(), C_, float(npix), 0)
    dx = torch.zeros(N, H, W, C_, dtype=torch.float16, device="cuda")
    eng.call("dy_bn_act_bwd_apply", dya.ptr, dya.ld, xa.ptr, xa.ld, dx.data_ptr(), C_, coef.data_ptr(), bwd.data_ptr(), npix, C_, act, 0)
    torch.cuda.synchronize()
    assert relerr(dg, gr.grad) < 1e-3 and relerr(db, br.grad) < 1e-3
    assert relerr(dx.float().permute(0, 3, 1, 2), xr.grad) < 3e-3


@pytest.mark.parametrize("cin,cout,ks,s,H,W,act", [(16, 32, 3, 1, 37, 45, 1), (64, 64, 3, 1, 40, 40, 1), (48, 32, 1, 1, 33, 29, 1),
                                                    (32, 64, 3, 2, 26, 34, 1), (128, 128, 1, 1, 20, 20, 2), (16, 16, 3, 1, 80, 80, 0)])
def test_bn_accumulator_path_equals_finalize_path(cin, cout, ks, s, H, W, act):
    """Round 3: statistics through fp64 accumulators (conv epilogue / backward reduce add, the apply kernels finish them in their
    prologue; no finalize launches) against the three-launch form on the same data: coefficients, running statistics, outputs,
    parameter gradients and dX must agree to fp32 rounding of the final division (the sums are the same fp32 partials added in
    fp64 in another order)."""
    from ultralytics.hip import DY_BN_COPIES
    torch.manual_seed(cin + cout + ks + act)
    eng = _eng()
    N = 4
    x = h16(torch.randn(N, cin, H, W))
    w = h16(torch.randn(cout, cin, ks, ks) / (cin * ks * ks) ** 0.5)
    sp = _spec(eng, w, None, ks, s)
    xa = _act(eng, x)
    Ho, Wo = eng.out_hw(sp, xa)
    npix = N * Ho * Wo
    gamma, beta = (torch.rand(cout) + 0.5).cuda(), (torch.randn(cout) * 0.2).cuda()
    dy = _act(eng, h16(torch.randn(N, cout, Ho, Wo)))

    def fwd_bwd(acc):
        rm, rv = torch.zeros(cout, device="cuda"), torch.ones(cout, device="cuda")
        coef = torch.zeros(4 * cout, device="cuda")
        raw = torch.zeros(N, Ho, Wo, cout, dtype=torch.float16, device="cuda")
        y = torch.zeros_like(raw)
        dx = torch.zeros_like(raw)
        dg, db = torch.zeros(cout, device="cuda"), torch.zeros(cout, device="cuda")
        if acc:
            af = torch.zeros(DY_BN_COPIES * 2 * cout, dtype=torch.float64, device="cuda")
            ab = torch.zeros_like(af)
            eng.call("dy_conv_forward", xa.ptr, xa.ld, sp.wpack.data_ptr(), 0, raw.data_ptr(), cout, af.data_ptr(), N, H, W, xa.C,
                     cout, ks, s, 1, 0, 0, 1 | 32, None)
            eng.call("dy_bn_act_apply_acc", raw.data_ptr(), cout, 0, 0, y.data_ptr(), cout, af.data_ptr(), gamma.data_ptr(),
                     beta.data_ptr(), rm.data_ptr(), rv.data_ptr(), coef.data_ptr(), npix, cout, act, float(npix), 1e-3, 0.03)
            eng.call("dy_bn_act_bwd_reduce_acc", dy.ptr, dy.ld, raw.data_ptr(), cout, coef.data_ptr(), ab.data_ptr(), npix, cout, act, 0, 0, 0)
            eng.call("dy_bn_act_bwd_apply_acc", dy.ptr, dy.ld, raw.data_ptr(), cout, dx.data_ptr(), cout, coef.data_ptr(),
                     ab.data_ptr(), dg.data_ptr(), db.data_ptr(), npix, cout, act, float(npix))
        else:
            nparts = eng.L.dy_conv_num_partials(N, H, W, xa.C, cout, ks, s, 1)
            part = torch.zeros(nparts, 2, cout, device="cuda")
            eng.call("dy_conv_forward", xa.ptr, xa.ld, sp.wpack.data_ptr(), 0, raw.data_ptr(), cout, part.data_ptr(), N, H, W, xa.C,
                     cout, ks, s, 1, 0, 0, 1, None)
            eng.call("dy_bn_finalize", part.data_ptr(), nparts, 1.0, 0, 0, 0.0, 0, 0, 0.0, gamma.data_ptr(), beta.data_ptr(),
                     rm.data_ptr(), rv.data_ptr(), coef.data_ptr(), cout, float(npix), 1e-3, 0.03, 1)
            eng.call("dy_bn_act_apply", raw.data_ptr(), cout, 0, 0, y.data_ptr(), cout, coef.data_ptr(), npix, cout, act)
            parts = torch.zeros(2048 * 2 * cout, device="cuda")
            n = C.c_int(0)
            eng.call("dy_bn_act_bwd_reduce", dy.ptr, dy.ld, raw.data_ptr(), cout, coef.data_ptr(), parts.data_ptr(), 2048, npix, cout,
                     act, C.byref(n))
            bwd = torch.zeros(2 * cout, device="cuda")
            eng.call("dy_bn_bwd_finalize", parts.data_ptr(), n.value, dg.data_ptr(), db.data_ptr(), bwd.data_ptr(), cout, float(npix), 0)
            eng.call("dy_bn_act_bwd_apply", dy.ptr, dy.ld, raw.data_ptr(), cout, dx.data_ptr(), cout, coef.data_ptr(), bwd.data_ptr(),
                     npix, cout, act, 0)
        torch.cuda.synchronize()
        return dict(coef=coef, rm=rm, rv=rv, raw=raw, y=y, dx=dx, dg=dg, db=db)

    a, b = fwd_bwd(True), fwd_bwd(False)
    assert torch.equal(a["raw"], b["raw"])
    # the backward apply pass folded into the weight-gradient kernel (dy_conv_wgrad_bn): d(raw) bit for bit what the stand-alone
    # pass writes, the same weight / input gradients as that pass followed by dy_conv_wgrad, and the BatchNorm parameter gradients
    from types import SimpleNamespace as NS
    af = torch.zeros(DY_BN_COPIES * 2 * cout, dtype=torch.float64, device="cuda")
    eng.call("dy_bn_act_bwd_reduce_acc", dy.ptr, dy.ld, a["raw"].data_ptr(), cout, a["coef"].data_ptr(), af.data_ptr(), npix, cout, act, 0, 0, 0)
    if act == 1:
        xa.needs_grad = True
        ref_dx = a["dx"].clone()
        eng._conv_bwd(sp, xa, ref_dx.data_ptr(), cout, Ho, Wo)
        torch.cuda.synchronize()
        gw_ref, gx_ref = sp.gweight.clone(), xa.st.gbuf.clone()
        sp.gweight.zero_(); xa.st.gbuf.zero_(); xa.st.gwritten.clear()
        draw = torch.full((N, Ho, Wo, cout), float("nan"), dtype=torch.float16, device="cuda")
        dg, db = torch.zeros(cout, device="cuda"), torch.zeros(cout, device="cuda")
        eng._conv_bwd(sp, xa, dy.ptr, dy.ld, Ho, Wo, bn=(NS(ptr=a["raw"].data_ptr(), ld=cout), draw, a["coef"].data_ptr(), af.data_ptr(),
                                                       dg.data_ptr(), db.data_ptr(), float(npix)))
        torch.cuda.synchronize()
        assert torch.equal(draw, ref_dx), float((draw.float() - ref_dx.float()).abs().max())
        assert torch.equal(sp.gweight, gw_ref) and torch.equal(xa.st.gbuf, gx_ref)
        assert torch.equal(dg, a["dg"]) and torch.equal(db, a["db"])
    for k in ("coef", "rm", "rv", "dg", "db"):
        assert relerr(a[k], b[k]) < 2e-6, k
    # fp16 outputs: identical except where a last-bit difference of a coefficient crosses an fp16 rounding boundary
    for k in ("y", "dx"):
        d = (a[k].float() - b[k].float()).abs()
        assert float((d > 0).float().mean()) < 1e-3 and relerr(a[k].float(), b[k].float()) < 1e-3, k
    # and the accumulator path repeats bit for bit (fp32 partials added in fp64: the order of the atomics does not show)
    a2 = fwd_bwd(True)
    for k in a:
        assert torch.equal(a[k], a2[k]), k


def test_pool_upsample_add():
    torch.manual_seed(5)
    eng = _eng()
    x = h16(torch.randn(2, 16, 11, 13))
    xa = _act(eng, x)
    eng.tape = []
    out = eng.new_act(2, 11, 13, 16)
    eng.maxpool5(xa, out)
    up = eng.upsample2x(out)
    s = eng.add([up, up])
    gy = h16(torch.randn(2, 16, 22, 26))
    s._gbuf()[...] = gy.cuda().permute(0, 2, 3, 1)
    s.st.gwritten.append((0, 16))
    for f in reversed(eng.tape):
        f()
    torch.cuda.synchronize()
    xr = x.clone().requires_grad_(True)
    p = F.max_pool2d(xr, 5, 1, 2)
    u = F.interpolate(p, scale_factor=2.0, mode="nearest")
    (u + u).backward(gy)
    assert relerr(s.st.buf.float().permute(0, 3, 1, 2), u + u) < 1e-3
    assert relerr(xa.st.gbuf.float().permute(0, 3, 1, 2), xr.grad) < 4e-3


@pytest.mark.parametrize("C,H,W,ties", [(64, 40, 40, False), (16, 11, 13, True), (8, 56, 56, False), (24, 20, 20, True)])
def test_sppf_fused_pools_equal_three_maxpool_launches(C, H, W, ties):
    """Engine.sppf_pools (one launch, the map in LDS, separable passes) against the three dy_maxpool5 launches it replaces: pooled
    slices, arg-max maps and the chained backward bit for bit -- with ties (values on a coarse grid, so that "first maximum in
    row-major window order" decides) and with a NaN in the map (reference nn/modules/block.py:166-171 through ATen's max_pool2d)."""
    import ultralytics.hip.engine as E
    torch.manual_seed(9)
    x = h16(torch.randn(3, C, H, W))
    if ties:
        x = (x * 2).round() / 2
        x[1, 3, H // 2, W // 2] = float("nan")
    g = h16(torch.randn(3, 4 * C, H, W))

    def run(fused):
        old, E.SPPF_FUSED = E.SPPF_FUSED, fused
        try:
            eng = _eng()
            eng.tape = []
            cat = eng.new_storage(3, H, W, 4 * C)
            cat.buf.zero_()
            cat.buf[..., :C] = x.cuda().permute(0, 2, 3, 1)
            assert bool(eng.L.dy_sppf_pool3_supported(H, W, C))
            n0 = len(eng.tape)
            eng.sppf_pools(cat, C)
            assert len(eng.tape) - n0 == (1 if fused else 3)
            a = cat.act()
            a._gbuf()[...] = g.cuda().permute(0, 2, 3, 1)
            cat.gwritten.append((0, 4 * C))
            for f in reversed(eng.tape):
                f()
            torch.cuda.synchronize()
            return cat.buf.clone(), cat.gbuf[..., :C].clone()
        finally:
            E.SPPF_FUSED = old

    y1, g1 = run(True)
    y0, g0 = run(False)
    assert torch.equal(torch.nan_to_num(y1.float(), nan=7e4), torch.nan_to_num(y0.float(), nan=7e4))
    dg = (torch.nan_to_num(g1.float(), nan=7e4) - torch.nan_to_num(g0.float(), nan=7e4)).abs()
    assert float(dg.max()) == 0.0, (float(dg.max()), int((dg > 0).sum()), (dg > 0).nonzero()[:4].tolist(), int(torch.isnan(g0).sum()), int(torch.isnan(g1).sum()))
    if not ties:  # and both are ATen's chain
        xr = x.clone().requires_grad_(True)
        ys = [xr]
        for _ in range(3):
            ys.append(F.max_pool2d(ys[-1], 5, 1, 2))
        cat = torch.cat(ys, 1)
        cat.backward(g)
        assert torch.equal(y1.float().permute(0, 3, 1, 2).cpu(), cat.detach())
        assert relerr(g1.float().permute(0, 3, 1, 2).cpu(), xr.grad) < 2e-3


def test_wgrad_deferred_batched_reduce_matches_immediate():
    """dy_conv_wgrad with dw == NULL + dy_wgrad_reduce_batched (what StepPlan's backward trace uses: every layer's slabs
    reduced by one launch) gives the same fp32 gradients as the immediate per-layer reduction."""
    eng = _eng()
    torch.manual_seed(3)
    specs, xs, dys = [], [], []
    for cin, cout, ks, s, H, W in [(16, 32, 3, 1, 12, 20), (32, 48, 1, 1, 9, 33), (16, 32, 3, 2, 16, 16)]:
        w = h16(torch.randn(cout, cin, ks, ks) / (cin * ks * ks) ** 0.5)
        sp = _spec(eng, w, None, ks, s)
        xa = _act(eng, h16(torch.randn(2, cin, H, W)))
        xa.needs_grad = False
        Ho, Wo = eng.out_hw(sp, xa)
        specs.append(sp); xs.append(xa); dys.append((_act(eng, h16(torch.randn(2, cout, Ho, Wo))), Ho, Wo))
    ref = []
    for sp, xa, (dya, Ho, Wo) in zip(specs, xs, dys):
        eng._conv_bwd(sp, xa, dya.ptr, dya.ld, Ho, Wo)
        torch.cuda.synchronize()
        ref.append(sp.gweight.clone())
        sp.gweight.zero_()
    eng.deferred_wgrad = []
    for sp, xa, (dya, Ho, Wo) in zip(specs, xs, dys):
        eng._conv_bwd(sp, xa, dya.ptr, dya.ld, Ho, Wo)
    assert len(eng.deferred_wgrad) == 3
    eng.flush_wgrad()
    torch.cuda.synchronize()
    assert eng.deferred_wgrad is None
    for sp, r in zip(specs, ref):
        assert torch.equal(sp.gweight, r)  # same slabs, same summation order


@pytest.mark.parametrize("H,W,Cc", [(16, 24, 16), (16, 32, 32), (8, 64, 16), (12, 16, 64)])
def test_scalseq_backward_all_levels_matches_per_level(H, W, Cc):
    """dy_scalseq_tail_backward_all (one pass per mode) against the per-level entry point it replaced in the engine.  (16, 24, 16) takes
    the block-per-lane kernel (24 columns are no multiple of 32), the other shapes the column-per-lane one."""
    import ctypes as C
    eng = _eng()
    L = eng.L
    torch.manual_seed(5)
    N = 2
    s = torch.cuda.current_stream().cuda_stream
    r = [torch.randn(N, H >> l, W >> l, Cc, device="cuda").half() for l in range(3)]
    dy = torch.randn(N, H, W, Cc, device="cuda").half()
    coef = torch.stack([torch.rand(Cc) + 0.5, torch.randn(Cc) * 0.1, torch.randn(Cc) * 0.1, torch.rand(Cc) + 0.5]).cuda().contiguous()
    rargs = (r[0].data_ptr(), Cc, r[1].data_ptr(), Cc, r[2].data_ptr(), Cc, dy.data_ptr(), Cc)
    # mode 0: BN partial sums
    part = torch.zeros(3 * 1024 * 2 * Cc, device="cuda")
    tot = 0
    for l in range(3):
        n = C.c_int(0)
        assert L.dy_scalseq_tail_backward(*rargs, 0, 0, coef.data_ptr(), 0, part.data_ptr() + tot * 2 * Cc * 4, 1024, N, H, W, Cc, l, 0,
                                          C.byref(n), s) == 0
        tot += n.value
    torch.cuda.synchronize()
    ref_sum = part[: tot * 2 * Cc].view(tot, 2, Cc).sum(0)
    part2 = torch.zeros(2048 * 2 * Cc, device="cuda")
    n = C.c_int(0)
    assert L.dy_scalseq_tail_backward_all(*rargs, 0, 0, 0, 0, 0, 0, coef.data_ptr(), 0, part2.data_ptr(), 2048, N, H, W, Cc, 0,
                                          C.byref(n), s) == 0
    torch.cuda.synchronize()
    got_sum = part2[: n.value * 2 * Cc].view(n.value, 2, Cc).sum(0)
    assert relerr(got_sum, ref_sum) < 1e-5
    # mode 1: dr0/dr1/dr2
    bw = (ref_sum / (3 * N * H * W)).contiguous()
    refs = []
    for l in range(3):
        d = torch.zeros(N, H >> l, W >> l, Cc, dtype=torch.float16, device="cuda")
        assert L.dy_scalseq_tail_backward(*rargs, d.data_ptr(), Cc, coef.data_ptr(), bw.data_ptr(), 0, 0, N, H, W, Cc, l, 1, None, s) == 0
        refs.append(d)
    outs = [torch.zeros_like(d) for d in refs]
    assert L.dy_scalseq_tail_backward_all(*rargs, outs[0].data_ptr(), Cc, outs[1].data_ptr(), Cc, outs[2].data_ptr(), Cc, coef.data_ptr(),
                                          bw.data_ptr(), 0, 0, N, H, W, Cc, 1, None, s) == 0
    torch.cuda.synchronize()
    for l in range(3):
        assert relerr(outs[l].float(), refs[l].float()) < 2e-3, l


@pytest.mark.parametrize("name", ["SGD", "Adam", "AdamW", "RMSProp", "RAdam", "Adamax", "NAdam"])
def test_flat_optimizer_step_matches_torch_optim(name):
    """dy_optimizer_step (unscale, global-norm clip, step per parameter group) against torch.optim on the CPU, built like the
    reference's build_optimizer (engine/trainer.py:1146-1180): betas = (momentum, 0.999), weight decay on the middle group."""
    import ctypes as C
    from ultralytics.hip import check, lib
    g = torch.Generator().manual_seed(5)
    n, g0, g1 = 3000, 500, 2200  # [0,g0) bias group, [g0,g1) decayed weights, rest norm weights
    p0 = torch.randn(n, generator=g)
    lr, mom, wd, scale, max_norm = 0.01, 0.9, 0.05, 8.0, 10.0
    ref_p = [p0[:g0].clone().requires_grad_(True), p0[g0:g1].clone().requires_grad_(True), p0[g1:].clone().requires_grad_(True)]
    groups = [dict(params=[ref_p[0]], weight_decay=0.0), dict(params=[ref_p[1]], weight_decay=wd), dict(params=[ref_p[2]], weight_decay=0.0)]
    opt = {"SGD": lambda: torch.optim.SGD(groups, lr=lr, momentum=mom, nesterov=True),
           "Adam": lambda: torch.optim.Adam(groups, lr=lr, betas=(mom, 0.999)),
           "AdamW": lambda: torch.optim.AdamW(groups, lr=lr, betas=(mom, 0.999)),
           "RMSProp": lambda: torch.optim.RMSprop(groups, lr=lr, momentum=mom),
           "RAdam": lambda: torch.optim.RAdam(groups, lr=lr, betas=(mom, 0.999)),
           "Adamax": lambda: torch.optim.Adamax(groups, lr=lr, betas=(mom, 0.999)),
           "NAdam": lambda: torch.optim.NAdam(groups, lr=lr, betas=(mom, 0.999))}[name]()
    mode = {"SGD": 0, "Adam": 1, "AdamW": 2, "RMSProp": 3, "RAdam": 4, "Adamax": 5, "NAdam": 6}[name]
    dev = "cuda:0"
    p = p0.clone().to(dev)
    m, v = torch.zeros(n, device=dev), torch.zeros(n, device=dev)
    hyper = torch.zeros(16, device=dev)
    hyper[0:3] = lr
    hyper[3], hyper[5], hyper[7], hyper[8], hyper[9], hyper[10] = mom, wd, 0.0, max_norm, 0.999, 1e-8
    state = torch.zeros(8, device=dev)
    state[0] = scale
    partials = torch.zeros(4096, device=dev)
    buf = torch.zeros(8, device=dev)
    for step in range(8):  # RAdam switches to the rectified update at step 6
        grad = torch.randn(n, generator=g) * (3.0 if step == 2 else 0.5)  # step 2 exceeds the clip norm
        for q, sl in zip(ref_p, (slice(0, g0), slice(g0, g1), slice(g1, n))):
            q.grad = grad[sl].clone()
        torch.nn.utils.clip_grad_norm_(ref_p, max_norm)
        opt.step()
        gd = (grad * scale).to(dev)
        check(lib().dy_optimizer_step(p.data_ptr(), gd.data_ptr(), m.data_ptr(), v.data_ptr(), None, n, g0, g1, None, buf.data_ptr(), None, 0,
                                      hyper.data_ptr(), state.data_ptr(), partials.data_ptr(), mode, None), "dy_optimizer_step")
        torch.cuda.synchronize()
        ref = torch.cat([q.detach() for q in ref_p])
        assert relerr(p, ref) < 2e-5, (name, step)
    assert float(state[5]) == 8 and float(state[6]) == 0


@pytest.mark.parametrize("N,H,W", [(2, 64, 96), (3, 70, 50), (1, 640, 640)])
def test_direct_stem_equals_import_plus_generic_conv(N, H, W):
    """model.0 = Conv(3 -> 16, k 3, s 2) read straight from the fp32 NCHW image batch (csrc/stem.hip: dy_stem_forward /
    dy_stem_wgrad_bn) against the three-launch form it replaces (import to fp16 NHWC8 + generic conv + generic fused weight
    gradient): raw output, BatchNorm sums, weight and BatchNorm parameter gradients."""
    from ultralytics.hip import DY_BN_COPIES
    torch.manual_seed(N * 1000 + H)
    eng = _eng()
    img = torch.rand(N, 3, H, W).cuda()
    w = h16(torch.randn(16, 3, 3, 3) / 27 ** 0.5)
    sp = _spec(eng, w, None, 3, 2)
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    npix = N * Ho * Wo
    xa = eng.import_image(img, 8, 1.0)
    gamma, beta = (torch.rand(16) + 0.5).cuda(), (torch.randn(16) * 0.2).cuda()
    dy = _act(eng, h16(torch.randn(N, 16, Ho, Wo)))

    def run(direct):
        raw = torch.zeros(N, Ho, Wo, 16, dtype=torch.float16, device="cuda")
        y = torch.zeros_like(raw)
        af = torch.zeros(DY_BN_COPIES * 2 * 16, dtype=torch.float64, device="cuda")
        ab = torch.zeros_like(af)
        rm, rv, coef = torch.zeros(16, device="cuda"), torch.ones(16, device="cuda"), torch.zeros(64, device="cuda")
        dg, db = torch.zeros(16, device="cuda"), torch.zeros(16, device="cuda")
        if direct:
            eng.call("dy_stem_forward", img.data_ptr(), sp.weight.data_ptr(), raw.data_ptr(), 16, af.data_ptr(), N, H, W, 1.0)
        else:
            eng.call("dy_conv_forward", xa.ptr, xa.ld, sp.wpack.data_ptr(), 0, raw.data_ptr(), 16, af.data_ptr(), N, H, W, xa.C, 16, 3, 2, 1,
                     0, 0, 1 | 32, None)
        eng.call("dy_bn_act_apply_acc", raw.data_ptr(), 16, 0, 0, y.data_ptr(), 16, af.data_ptr(), gamma.data_ptr(), beta.data_ptr(),
                 rm.data_ptr(), rv.data_ptr(), coef.data_ptr(), npix, 16, 1, float(npix), 1e-3, 0.03)
        eng.call("dy_bn_act_bwd_reduce_acc", dy.ptr, dy.ld, raw.data_ptr(), 16, coef.data_ptr(), ab.data_ptr(), npix, 16, 1, 0, 0, 0)
        sp.gweight.zero_()
        if direct:
            ns = eng.L.dy_stem_grid(N, H, W)
            slabs = torch.full((ns * 9 * 16 * 16,), float("nan"), device="cuda")
            eng.call("dy_stem_wgrad_bn", img.data_ptr(), dy.ptr, dy.ld, raw.data_ptr(), 16, coef.data_ptr(), ab.data_ptr(), dg.data_ptr(),
                     db.data_ptr(), float(npix), slabs.data_ptr(), N, H, W, 1.0)
            eng.deferred_wgrad = [(sp, slabs, ns)]
            eng.flush_wgrad()
        else:
            from types import SimpleNamespace as NS
            xa.needs_grad = False
            eng._conv_bwd(sp, xa, dy.ptr, dy.ld, Ho, Wo, bn=(NS(ptr=raw.data_ptr(), ld=16), raw, coef.data_ptr(), ab.data_ptr(), dg.data_ptr(),
                                                           db.data_ptr(), float(npix)))
        torch.cuda.synchronize()
        sums = af.view(DY_BN_COPIES, 2, 16).sum(0)
        return dict(raw=raw.float(), sums=sums, y=y.float(), gw=sp.gweight.clone(), dg=dg, db=db)

    a, b = run(True), run(False)
    ref = F.conv2d(img.half().float().cpu(), w, None, 2, 1).permute(0, 2, 3, 1)
    assert relerr(a["raw"], ref) < 1e-3 and relerr(b["raw"], ref) < 1e-3
    assert relerr(a["raw"], b["raw"]) < 1e-3 and float((a["raw"] != b["raw"]).float().mean()) < 2e-2   # summation order only
    assert relerr(a["sums"], b["sums"]) < 1e-5
    assert relerr(a["gw"], b["gw"]) < 2e-3 and relerr(a["dg"], b["dg"]) < 2e-3 and relerr(a["db"], b["db"]) < 2e-3


@pytest.mark.parametrize("cp,cq,ks,H,W", [(32, 32, 1, 37, 45), (16, 16, 3, 40, 40), (32, 32, 3, 33, 35), (64, 64, 3, 24, 40), (64, 64, 1, 31, 17),
                                          (32, 8, 1, 20, 20), (128, 64, 1, 10, 10)])
def test_bn_backward_reduce_in_the_dgrad_epilogue(cp, cq, ks, H, W):
    """dy_conv_input_grad_red: the stride-1 input gradient that is the only writer of a Conv's output gradient also runs the first pass
    of that Conv's BatchNorm backward in its epilogue.  Against the two launches it replaces (dy_conv_forward over the transposed pack,
    then dy_bn_act_bwd_reduce_acc on the stored gradient): the same gradient tensor bit for bit, the same sums up to fp32 summation order."""
    from ultralytics.hip import DY_BN_COPIES, check
    torch.manual_seed(cp + cq + ks)
    eng = _eng()
    N = 3
    if not eng.L.dy_conv_red_supported((cq + 7) // 8 * 8, cp, ks):
        pytest.skip("geometry outside the ping-pong kernel")
    w = h16(torch.randn(cq, cp, ks, ks) / (cp * ks * ks) ** 0.5)   # the consumer conv: cp -> cq
    sp = _spec(eng, w, None, ks, 1)
    dy = _act(eng, h16(torch.randn(N, cq, H, W)))                  # gradient w.r.t. the consumer's raw output
    raw = h16(torch.randn(N, H, W, cp) * 1.5 + 0.2).half().cuda().contiguous()   # the producer's raw conv output
    coef = torch.cat([torch.rand(cp) + 0.5, torch.randn(cp) * 0.3, torch.randn(cp) * 0.2, torch.rand(cp) + 0.5]).cuda()
    s = torch.cuda.current_stream().cuda_stream
    dx_a = torch.full((N, H, W, cp), float("nan"), dtype=torch.float16, device="cuda")
    acc_a = torch.zeros(DY_BN_COPIES * 2 * cp, dtype=torch.float64, device="cuda")
    check(eng.L.dy_conv_input_grad_red(dy.ptr, dy.ld, sp.wpack_t.data_ptr(), dx_a.data_ptr(), cp, N, H, W, sp.cout_phys, cp, ks, raw.data_ptr(), cp,
                                       coef.data_ptr(), acc_a.data_ptr(), cp, s), "dy_conv_input_grad_red")
    dx_b = torch.full_like(dx_a, float("nan"))
    acc_b = torch.zeros_like(acc_a)
    eng.call("dy_conv_forward", dy.ptr, dy.ld, sp.wpack_t.data_ptr(), 0, dx_b.data_ptr(), cp, 0, N, H, W, sp.cout_phys, cp, ks, 1, 1, 0, 0, 0, None)
    eng.call("dy_bn_act_bwd_reduce_acc", dx_b.data_ptr(), cp, raw.data_ptr(), cp, coef.data_ptr(), acc_b.data_ptr(), N * H * W, cp, 1, 0, 0, 0)
    torch.cuda.synchronize()
    assert torch.equal(dx_a, dx_b)
    sa, sb = acc_a.view(DY_BN_COPIES, 2, cp).sum(0), acc_b.view(DY_BN_COPIES, 2, cp).sum(0)
    assert relerr(sa, sb) < 2e-5, relerr(sa, sb)


@pytest.mark.parametrize("ratio", [1.0, 30.0, 300.0])
def test_bn_backward_reduce_with_a_large_mean_to_std_ratio(ratio):
    """The backward reduce sums g and g*x on the RAW conv outputs and forms sum(g * xhat) = (sum(g x) - mean sum(g)) * invstd
    afterwards (per thread, fp32; blocks are added in fp64): a cancellation when |mean| >> std.  Measured against an fp64
    evaluation of the same sums on the same fp16 inputs -- the bound grows with the ratio as eps_fp32 * ratio * sqrt(terms)
    and stays far below what the fp16 storage of the activations itself costs."""
    torch.manual_seed(11)
    eng = _eng()
    N, C, H, W = 4, 32, 40, 40
    std = 0.5
    x = h16(torch.randn(N, C, H, W) * std + ratio * std)
    dy = h16(torch.randn(N, C, H, W))
    xa, da = _act(eng, x), _act(eng, dy)
    npix = N * H * W
    xf = x.double().permute(0, 2, 3, 1).reshape(npix, C)
    mean = xf.mean(0)
    var = xf.var(0, unbiased=False)
    inv = 1.0 / torch.sqrt(var + 1e-3)
    gamma = torch.rand(C, dtype=torch.float64) + 0.5
    beta = torch.randn(C, dtype=torch.float64) * 0.1
    coef = torch.stack([gamma * inv, beta - mean * gamma * inv, mean, inv]).float().cuda().contiguous()
    from ultralytics.hip import DY_BN_COPIES, check
    acc = torch.zeros(DY_BN_COPIES, 2, C, dtype=torch.float64, device="cuda")
    check(eng.L.dy_bn_act_bwd_reduce_acc(da.ptr, da.ld, xa.ptr, xa.ld, coef.data_ptr(), acc.data_ptr(), npix, C, 1, 0, 0, 0, eng.stream), "reduce")
    torch.cuda.synchronize()
    got = acc.sum(0).cpu()
    # fp64 reference with the coefficients the kernel saw (fp32 scale / shift / mean / invstd)
    cf = coef.double().cpu()
    z = xf * cf[0] + cf[1]
    s = torch.sigmoid(z)
    g = dy.double().permute(0, 2, 3, 1).reshape(npix, C) * (s * (1 + z * (1 - s)))
    ref = torch.stack([g.sum(0), (g * (xf - cf[2]) * cf[3]).sum(0)])
    scale = torch.stack([g.abs().sum(0), (g * (xf - cf[2]) * cf[3]).abs().sum(0)])
    err = float(((got - ref).abs() / scale).max())
    print(f"mean/std {ratio:g}: worst |error| / sum|terms| = {err:.2e}")
    # measured 1.1e-8 / 1.1e-7 / 1.1e-6 (linear in the ratio, as the fp32 cancellation predicts); fp16 storage of x alone costs
    # 5e-4 relative per element
    assert err < {1.0: 1e-7, 30.0: 1e-6, 300.0: 1e-5}[ratio]


@pytest.mark.parametrize("acc", [0, 1])
def test_conv1x1_rows_backward_reads_foreground_rows_only(acc):
    """dy_conv1x1_rows_backward (Detect's final box conv from the rows of its gradient, reference nn/modules/head.py:38-40 +
    utils/loss.py:436-445): weight / bias / input gradients against the dense fp32 products, with NaN in every background row of the
    gradient -- a background row that is read shows up in all three."""
    import ctypes as C
    from ultralytics.hip import DY_BN_COPIES, check, lib
    L = lib()
    torch.manual_seed(4)
    B, H, W, A0, Aextra = 3, 9, 13, 50, 7
    hw, A = H * W, 50 + 9 * 13 + 7
    npix = B * hw
    x = h16(torch.randn(npix, 64)).cuda().half()
    w = (torch.randn(64, 64) / 8).cuda()
    fg = torch.rand(B, hw) < 0.08
    fg[0, 0] = fg[2, hw - 1] = True
    asg = torch.full((B, A), -1, dtype=torch.int32)
    asg[:, :A0] = 3  # other levels' anchors: foreground there must not leak into this level
    asg[:, A0:A0 + hw][fg] = 1
    dy = torch.full((npix, 64), float("nan"), dtype=torch.float16)
    rows = fg.reshape(-1)
    dy[rows] = h16(torch.randn(int(rows.sum()), 64)).half()
    dyc = dy.cuda()
    dx0 = h16(torch.randn(npix, 64)).cuda().half()
    dx = dx0.clone()
    ns = L.dy_conv1x1_rows_slabs(B, H, W)
    slabs = torch.full((ns, 64, 64), float("nan"), device="cuda")
    bacc = torch.zeros(DY_BN_COPIES, 64, dtype=torch.float64, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    check(L.dy_conv1x1_rows_backward(x.data_ptr(), 64, 0, dyc.data_ptr(), 64, asg.cuda().data_ptr(), A, A0, w.data_ptr(), dx.data_ptr(), 64, acc,
                                     slabs.data_ptr(), bacc.data_ptr(), B, H, W, 64, 64, s), "rows")
    torch.cuda.synchronize()
    d = torch.nan_to_num(dy.float(), nan=0.0).cuda()
    dw_ref = d.t().double() @ x.double()
    assert relerr(slabs.double().sum(0), dw_ref) < 1e-6
    assert relerr(bacc.sum(0), d.double().sum(0)) < 1e-6
    dx_ref = d @ w.half().float() + (dx0.float() if acc else 0.0)
    assert torch.isfinite(dx.float()).all()
    assert relerr(dx.float(), dx_ref) < 2e-3
    bg = ~rows.cuda()
    assert torch.equal(dx[bg], dx0[bg] if acc else torch.zeros_like(dx0[bg]))


def test_bn_backward_reduce_rows_equals_the_dense_pass_on_a_row_sparse_gradient():
    """dy_bn_act_bwd_reduce_rows against dy_bn_act_bwd_reduce_acc on the same gradient (zero outside the foreground rows): the sums
    agree to fp32 summation order; background rows of dy hold NaN for the rows form -- it must not read them."""
    from ultralytics.hip import DY_BN_COPIES, check, lib
    L = lib()
    torch.manual_seed(6)
    B, H, W, C_, A0 = 4, 20, 20, 64, 37
    hw, A = H * W, 37 + 400 + 11
    npix = B * hw
    x = (torch.randn(npix, C_) * 0.7 + 0.2).half().cuda()
    fg = torch.rand(B, hw) < 0.05
    fg[1, 7] = True
    asg = torch.full((B, A), -1, dtype=torch.int32)
    asg[:, A0:A0 + hw][fg] = 0
    rows = fg.reshape(-1)
    dense = torch.zeros(npix, C_, dtype=torch.float16)
    dense[rows] = torch.randn(int(rows.sum()), C_).half()
    sparse = dense.clone()
    sparse[~rows] = float("nan")
    coef = torch.stack([torch.rand(C_) + 0.5, torch.randn(C_) * 0.1, torch.randn(C_) * 0.2, torch.rand(C_) + 0.5]).cuda().contiguous()
    s = torch.cuda.current_stream().cuda_stream
    acc_d = torch.zeros(DY_BN_COPIES, 2, C_, dtype=torch.float64, device="cuda")
    acc_r = torch.zeros_like(acc_d)
    dd, ds, asg_c = dense.cuda(), sparse.cuda(), asg.cuda()
    check(L.dy_bn_act_bwd_reduce_acc(dd.data_ptr(), C_, x.data_ptr(), C_, coef.data_ptr(), acc_d.data_ptr(), npix, C_, 1, 0, 0, 0, s), "dense")
    check(L.dy_bn_act_bwd_reduce_rows(ds.data_ptr(), C_, x.data_ptr(), C_, coef.data_ptr(), acc_r.data_ptr(), B, hw, C_, 1, asg_c.data_ptr(), A, A0, s), "rows")
    torch.cuda.synchronize()
    a, b = acc_d.sum(0), acc_r.sum(0)
    assert torch.isfinite(b).all() and relerr(b, a) < 2e-6


def test_head_box_decode_equals_conv_then_bbox_decode():
    """dy_head_box_decode (Detect's final box conv fused with the loss's bbox_decode, reference nn/modules/head.py:38-40 +
    utils/loss.py:347-354 + utils/tal.py:311-318 dist2bbox) against the fp32 composition: logits = x W^T + b with the fp16-rounded
    weight, softmax over the 16 bins of each side, expectation, anchor -/+ distance; ragged pixel count, anchors of other levels
    around this level's range."""
    from ultralytics.hip import check, lib
    L = lib()
    torch.manual_seed(8)
    B, H, W, A0 = 3, 7, 11, 19
    hw, A = H * W, 19 + 77 + 5
    x = torch.randn(B * hw, 64).half().cuda()
    w = (torch.randn(64, 64) / 4).cuda()
    b = torch.randn(64).cuda()
    pb = torch.full((B, A, 4), -7.0, device="cuda")
    check(L.dy_head_box_decode(x.data_ptr(), 64, 0, w.data_ptr(), b.data_ptr(), pb.data_ptr(), A, A0, B, H, W, 64, 64,
                               torch.cuda.current_stream().cuda_stream), "decode")
    torch.cuda.synchronize()
    logits = x.float() @ w.half().float().t() + b
    e = (torch.softmax(logits.view(B, hw, 4, 16), -1) * torch.arange(16, device="cuda")).sum(-1)
    iy, ix = torch.meshgrid(torch.arange(H, device="cuda"), torch.arange(W, device="cuda"), indexing="ij")
    anc = torch.stack([ix, iy], -1).reshape(hw, 2).float() + 0.5
    ref = torch.cat([anc - e[..., :2], anc + e[..., 2:]], -1)
    assert float((pb[:, A0:A0 + hw] - ref).abs().max()) < 2e-5
    assert bool((pb[:, :A0] == -7.0).all()) and bool((pb[:, A0 + hw:] == -7.0).all())
    # the same from the RAW output of the Conv in front + its coefficient table: what feeding the tensor the apply launch writes gives
    raw = torch.randn(B * hw, 64).half().cuda()
    coef = torch.stack([torch.rand(64) + 0.5, torch.randn(64) * 0.3, torch.zeros(64), torch.ones(64)]).cuda().contiguous()
    y = torch.empty_like(raw)
    s_ = torch.cuda.current_stream().cuda_stream
    check(L.dy_bn_act_apply(raw.data_ptr(), 64, 0, 0, y.data_ptr(), 64, coef.data_ptr(), B * hw, 64, 1, s_), "apply")
    pb1, pb2 = torch.zeros((B, A, 4), device="cuda"), torch.zeros((B, A, 4), device="cuda")
    check(L.dy_head_box_decode(y.data_ptr(), 64, 0, w.data_ptr(), b.data_ptr(), pb1.data_ptr(), A, A0, B, H, W, 64, 64, s_), "decode")
    check(L.dy_head_box_decode(raw.data_ptr(), 64, coef.data_ptr(), w.data_ptr(), b.data_ptr(), pb2.data_ptr(), A, A0, B, H, W, 64, 64, s_), "decode")
    torch.cuda.synchronize()
    d = float((pb1 - pb2).abs().max())
    print("decode from (raw, coef) vs from the applied tensor:", d)
    assert d < 2e-3  # the plain apply kernel (library division) vs the packed form: an fp16 output differs in its last bit now and then


@pytest.mark.parametrize("cin,nc,coef", [(32, 6, True), (64, 8, False), (32, 1, True)])
def test_cls_head_kernels_equal_the_dense_products(cin, nc, coef):
    """dy_cls_head_forward / dy_cls_head_backward (Detect's final class conv, reference nn/modules/head.py:41-42) against fp32 torch
    products: logits, weight / bias gradient (slabs + fp64 sums) and the input gradient, from an activated input and from the RAW
    output of the Conv in front + its coefficient table; ragged pixel count."""
    from ultralytics.hip import DY_BN_COPIES, check, lib
    L = lib()
    torch.manual_seed(12)
    npix = 3 * 37 * 23 + 5
    raw = torch.randn(npix, cin).half().cuda()
    w = (torch.randn(nc, cin) / 4).cuda()
    b = torch.randn(nc).cuda()
    s = torch.cuda.current_stream().cuda_stream
    if coef:
        cf = torch.stack([torch.rand(cin) + 0.5, torch.randn(cin) * 0.3, torch.zeros(cin), torch.ones(cin)]).cuda().contiguous()
        z = raw.float() * cf[0] + cf[1]
        x = (z * torch.sigmoid(z)).half()
        xp, cp = raw.data_ptr(), cf.data_ptr()
    else:
        x, xp, cp = raw, raw.data_ptr(), 0
    logits = torch.full((npix, 8), 3.0, device="cuda")
    check(L.dy_cls_head_forward(xp, cin, cp, w.data_ptr(), b.data_ptr(), logits.data_ptr(), npix, cin, nc, s), "fwd")
    ref = x.float() @ w.half().float().t() + b
    torch.cuda.synchronize()
    assert relerr(logits[:, :nc], ref) < 2e-3 and bool((logits[:, nc:] == 0).all())
    dy = torch.zeros(npix, 8, dtype=torch.float16, device="cuda")
    dy[:, :nc] = torch.randn(npix, nc).half()
    ns = L.dy_cls_head_slabs()
    slabs = torch.full((ns, 16, cin), float("nan"), device="cuda")
    bacc = torch.zeros(DY_BN_COPIES, 8, dtype=torch.float64, device="cuda")
    dx0 = torch.randn(npix, cin).half().cuda()
    for acc in (0, 1):
        dx = dx0.clone()
        bacc.zero_()
        check(L.dy_cls_head_backward(xp, cin, cp, dy.data_ptr(), w.data_ptr(), dx.data_ptr(), cin, acc, slabs.data_ptr(), bacc.data_ptr(),
                                     npix, cin, nc, s), "bwd")
        torch.cuda.synchronize()
        dw = slabs.sum(0)
        assert relerr(dw[:nc], dy[:, :nc].float().t() @ x.float()) < 2e-3 and bool((dw[nc:] == 0).all())
        assert relerr(bacc.sum(0)[:nc].float(), dy[:, :nc].float().sum(0)) < 1e-5
        dx_ref = dy[:, :nc].float() @ w.half().float() + (dx0.float() if acc else 0.0)
        assert relerr(dx.float(), dx_ref) < 2e-3
