"""-m gpu: the LDConv sampling backward through the C-ABI -- the deterministic gather (dy_ldconv_sample_backward_gather)
with its atomic side pass for samples beyond the gather radius, and the plain scatter entry -- against torch autograd over
the oracle's restatement of the reference's sampling stage (oracle.nn.ld_sample, nn/modules/conv.py:368-404)."""
import pytest
import torch

from gpu_util import h16, relerr

pytestmark = pytest.mark.gpu

# (C, Np, stride, H, W, offset magnitude): sub-pixel / typical trained range / beyond the gather radius (near samples
# gathered, far ones through the side pass) / far outside the map (every clamp branch); odd sizes, C with 1 / 2 / 4 granules
# per thread
CASES = [(16, 3, 2, 24, 36, 0.0), (16, 3, 2, 24, 36, 0.9), (32, 1, 1, 17, 23, 1.7), (64, 3, 2, 15, 21, 2.5), (128, 1, 1, 9, 12, 3.9),
         (8, 3, 2, 20, 28, 1.2), (24, 1, 1, 11, 13, 0.7), (32, 3, 2, 16, 16, 6.0), (16, 1, 1, 10, 14, 40.0), (32, 5, 1, 12, 10, 1.5)]


def _reference(x, off, gxo, Np, s):
    from oracle.graph import ld_p_n
    from oracle.nn import ld_sample
    x = x.clone().requires_grad_(True)
    off = off.clone().requires_grad_(True)
    xo = ld_sample(x, off, ld_p_n(Np), Np, s)  # (B,C,h,w,N)
    xo.backward(gxo)
    return xo.detach(), x.grad, off.grad


def _run(C_, Np, s, H, W, mag, mode, accumulate=0, rmax=2):
    from oracle.graph import ld_p_n
    from ultralytics.hip import check, lib
    g = torch.Generator().manual_seed(C_ * 1000 + Np * 100 + H)
    B = 2
    h, w = (H + s - 1) // s if s > 1 else H, (W + s - 1) // s if s > 1 else W
    x = h16(torch.randn(B, C_, H, W, generator=g))
    off = (torch.rand(B, 2 * Np, h, w, generator=g) * 2 - 1) * mag
    if mag > 0:
        off[0, :, 0, 0] = mag  # pin the layer maximum so that the radius is what the case says
    gxo = h16(torch.randn(B, C_, h, w, Np, generator=g))
    prev = h16(torch.randn(B, C_, H, W, generator=g)) if accumulate else torch.zeros(B, C_, H, W)
    xo_ref, gx_ref, goff_ref = _reference(x, off, gxo, Np, s)
    dev = "cuda:0"
    xd = x.permute(0, 2, 3, 1).contiguous().half().to(dev)
    offd = off.permute(0, 2, 3, 1).contiguous().to(dev)
    pn = ld_p_n(Np).reshape(-1).to(torch.int32).to(dev)
    xo = torch.empty(B, h, w, Np * C_, dtype=torch.float16, device=dev)
    check(lib().dy_ldconv_sample(xd.data_ptr(), C_, offd.data_ptr(), 2 * Np, pn.data_ptr(), xo.data_ptr(), Np * C_, B, H, W, h, w, C_, Np, s, None), "dy_ldconv_sample")
    # x_off[pix][n*C + c]
    gd = gxo.permute(0, 2, 3, 4, 1).reshape(B, h, w, Np * C_).contiguous().half().to(dev)
    lddoff = 8 * ((2 * Np + 7) // 8)
    doff = torch.zeros(B, h, w, lddoff, dtype=torch.float16, device=dev)
    dx = prev.permute(0, 2, 3, 1).contiguous().half().to(dev)
    dx32 = torch.full((B, H, W, C_), 7.0, device=dev)  # poisoned: the fallback must zero it itself
    scratch = torch.full((4,), 123, dtype=torch.int32, device=dev)
    if mode == "gather":
        check(lib().dy_ldconv_sample_backward_gather(xd.data_ptr(), C_, offd.data_ptr(), 2 * Np, pn.data_ptr(), gd.data_ptr(), Np * C_,
                                                     dx.data_ptr(), C_, accumulate, dx32.data_ptr(), doff.data_ptr(), lddoff,
                                                     scratch.data_ptr(), rmax, B, H, W, h, w, C_, Np, s, None), "dy_ldconv_sample_backward_gather")
        measured = scratch[:1].view(torch.float32).item()
        assert abs(measured - off.abs().max().item()) < 1e-6
    else:
        dx32.zero_()
        check(lib().dy_ldconv_sample_backward(xd.data_ptr(), C_, offd.data_ptr(), 2 * Np, pn.data_ptr(), gd.data_ptr(), Np * C_,
                                              dx32.data_ptr(), doff.data_ptr(), lddoff, B, H, W, h, w, C_, Np, s, None), "dy_ldconv_sample_backward")
        check(lib().dy_f32_to_f16_add(dx32.data_ptr(), dx.data_ptr(), C_, B * H * W, C_, accumulate, None), "dy_f32_to_f16_add")
    torch.cuda.synchronize()
    xo_g = xo.float().cpu().view(B, h, w, Np, C_).permute(0, 4, 1, 2, 3)
    gx = dx.float().cpu().permute(0, 3, 1, 2)
    goff = doff.float().cpu()[..., :2 * Np].permute(0, 3, 1, 2)
    return (xo_g, xo_ref), (gx, gx_ref + prev), (goff, goff_ref), dx


@pytest.mark.parametrize("rmax", [1, 2, 4])
@pytest.mark.parametrize("case", CASES, ids=lambda c: "C{}_N{}_s{}_{}x{}_off{}".format(*c))
def test_gather_backward_matches_autograd(case, rmax):
    (xo, xo_ref), (gx, gx_ref), (goff, goff_ref), _ = _run(*case, mode="gather", rmax=rmax)
    assert relerr(xo, xo_ref) < 2e-3
    assert relerr(gx, gx_ref) < 2e-3  # one fp16 rounding of an fp32 sum
    assert relerr(goff, goff_ref) < 3e-3


@pytest.mark.parametrize("case", [CASES[1], CASES[3], CASES[8]], ids=lambda c: "C{}_N{}_s{}_{}x{}_off{}".format(*c))
def test_gather_accumulates_and_agrees_with_scatter(case):
    (_, _), (gx, gx_ref), _, _ = _run(*case, mode="gather", accumulate=1)
    assert relerr(gx, gx_ref) < 2e-3
    (_, _), (gs, _), _, _ = _run(*case, mode="scatter", accumulate=1)
    assert relerr(gs, gx_ref) < 2e-3
    assert relerr(gx, gs) < 2e-3


def test_gather_is_bitwise_reproducible():
    """No far sample (|offset| <= 2.5 < rmax = 4): no atomics anywhere, so two runs agree bit for bit."""
    a = _run(*CASES[3], mode="gather", rmax=4)[3]
    b = _run(*CASES[3], mode="gather", rmax=4)[3]
    assert torch.equal(a, b)
