import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "experiment-yolo_amd")]
import torch, ctypes as C
from ultralytics.hip.engine import Engine
eng = Engine("cuda:0"); L = eng.L
s = torch.cuda.current_stream().cuda_stream
torch.manual_seed(0)
for npix, Cc, act in [(2048, 16, 1), (2048 * 3 + 77, 16, 1), (5000, 64, 1), (333, 48, 2), (100000, 32, 1), (4097, 8, 0)]:
    x = (torch.randn(npix, Cc, device="cuda") * 2 + 3).half()
    dy = torch.randn(npix, Cc, device="cuda").half()
    coef = torch.empty(4, Cc, device="cuda")
    mean = x.float().mean(0); var = x.float().var(0, unbiased=False); inv = (var + 1e-3).rsqrt()
    gamma = torch.rand(Cc, device="cuda") + 0.5; beta = torch.randn(Cc, device="cuda")
    coef[0] = gamma * inv; coef[1] = beta - mean * gamma * inv; coef[2] = mean; coef[3] = inv
    z = x.float() * coef[0] + coef[1]
    if act == 1:
        sg = torch.sigmoid(z); fwd = z * sg; gr = sg * (1 + z * (1 - sg))
    elif act == 2:
        fwd = torch.where(z > 0, z, 0.1 * z); gr = torch.where(z > 0, torch.ones_like(z), torch.full_like(z, 0.1))
    else:
        fwd = z; gr = torch.ones_like(z)
    y = torch.empty_like(x)
    L.dy_bn_act_apply(x.data_ptr(), Cc, 0, 0, y.data_ptr(), Cc, coef.data_ptr(), npix, Cc, act, s)
    part = torch.zeros(2048 * 2 * Cc, device="cuda"); n = C.c_int(0)
    L.dy_bn_act_bwd_reduce(dy.data_ptr(), Cc, x.data_ptr(), Cc, coef.data_ptr(), part.data_ptr(), 2048, npix, Cc, act, C.byref(n), s)
    torch.cuda.synchronize()
    P = part[: n.value * 2 * Cc].view(n.value, 2, Cc).sum(0)
    g = dy.float() * gr; xh = (x.float() - mean) * inv
    e_ap = (y.float() - fwd).abs().max() / fwd.abs().max()
    e_r1 = (P[0] - g.sum(0)).abs().max() / g.sum(0).abs().max()
    e_r2 = (P[1] - (g * xh).sum(0)).abs().max() / (g * xh).sum(0).abs().max()
    bw = torch.stack([g.mean(0), (g * xh).mean(0)]).contiguous()
    dx = torch.empty_like(x)
    L.dy_bn_act_bwd_apply(dy.data_ptr(), Cc, x.data_ptr(), Cc, dx.data_ptr(), Cc, coef.data_ptr(), bw.data_ptr(), npix, Cc, act, 0, s)
    torch.cuda.synchronize()
    ref = coef[0] * (g - bw[0] - xh * bw[1])
    e_ba = (dx.float() - ref).abs().max() / ref.abs().max()
    print(f"npix={npix} C={Cc} act={act}: apply {e_ap:.2e} reduce {e_r1:.2e} {e_r2:.2e} bwd_apply {e_ba:.2e} nparts {n.value}")
