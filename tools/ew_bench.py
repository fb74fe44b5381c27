#!/usr/bin/env python3
"""Micro-benchmark of the element-wise BN kernels (GB/s at their algorithmic bytes). usage: ew_bench.py [npix C]..."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "experiment-yolo_amd")]
import torch  # noqa: E402
import ctypes as C  # noqa: E402
from ultralytics.hip.engine import Engine  # noqa: E402

eng = Engine("cuda:0")
L = eng.L
shapes = [(1638400, 64), (6553600, 16), (1638400, 32), (409600, 64), (409600, 32), (102400, 64), (102400, 128)]


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for npix, Cc in shapes:
    x = torch.randn(npix, Cc, device="cuda").half()
    dy = torch.randn(npix, Cc, device="cuda").half()
    y = torch.empty_like(x)
    coef = torch.rand(4 * Cc, device="cuda") + 0.5
    bw = torch.rand(2 * Cc, device="cuda") * 0.01
    part = torch.zeros(2048 * 2 * Cc, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    n = C.c_int(0)
    t_ap = timeit(lambda: L.dy_bn_act_apply(x.data_ptr(), Cc, 0, 0, y.data_ptr(), Cc, coef.data_ptr(), npix, Cc, 1, s))
    t_rd = timeit(lambda: L.dy_bn_act_bwd_reduce(dy.data_ptr(), Cc, x.data_ptr(), Cc, coef.data_ptr(), part.data_ptr(), 2048, npix, Cc, 1, C.byref(n), s))
    t_ba = timeit(lambda: L.dy_bn_act_bwd_apply(dy.data_ptr(), Cc, x.data_ptr(), Cc, y.data_ptr(), Cc, coef.data_ptr(), bw.data_ptr(), npix, Cc, 1, 0, s))
    e = npix * Cc
    print(f"npix={npix} C={Cc}: apply {t_ap*1e3:6.1f} us {4*e/t_ap/1e6:5.0f} GB/s | bwd_reduce {t_rd*1e3:6.1f} us {4*e/t_rd/1e6:5.0f} GB/s | bwd_apply {t_ba*1e3:6.1f} us {6*e/t_ba/1e6:5.0f} GB/s")

# reference point: a plain device-to-device copy of the same element count (2 B read + 2 B write per element)
for npix, Cc in shapes[:2]:
    a = torch.randn(npix, Cc, device="cuda").half()
    b = torch.empty_like(a)
    t = timeit(lambda: b.copy_(a))
    print(f"copy npix={npix} C={Cc}: {t*1e3:6.1f} us {4*npix*Cc/t/1e6:5.0f} GB/s")
