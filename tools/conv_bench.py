#!/usr/bin/env python3
"""Micro-benchmark of single dy_conv_forward / dy_conv_wgrad launches (algorithmic GB/s and TFLOP/s).
usage: conv_bench.py [fwd|wgrad] cin cout ks stride H W [N] [reps]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "experiment-yolo_amd")]
import torch  # noqa: E402

from ultralytics.hip.engine import ConvSpec, Engine, Storage  # noqa: E402

mode, cin, cout, ks, s, H, W = sys.argv[1], *map(int, sys.argv[2:8])
N = int(sys.argv[8]) if len(sys.argv) > 8 else 64
reps = int(sys.argv[9]) if len(sys.argv) > 9 else 20
eng = Engine("cuda:0")
w = torch.randn(cout, cin, ks, ks, device="cuda") / (cin * ks * ks) ** 0.5
sp = ConvSpec("b", w, None, None, ks, s, 0)
sp.gweight = torch.zeros_like(w)
eng.prepare_conv(sp)
eng.pack(sp)
cp = (cin + 7) // 8 * 8
x = Storage(eng, N, H, W, cp)
x.buf.copy_(torch.randn_like(x.buf))
xa = x.act()
Ho, Wo = eng.out_hw(sp, xa)
y = Storage(eng, N, Ho, Wo, (cout + 7) // 8 * 8)
y.buf.copy_(torch.randn_like(y.buf))
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]


def run():
    if mode == "fwd":
        eng._conv_raw(sp, xa, y.buf.data_ptr(), y.C, int(os.environ.get("DY_EPI", "0")))
    else:
        eng._conv_bwd(sp, Storage.act(x) if False else xa, y.buf.data_ptr(), y.C, Ho, Wo)


xa.needs_grad = False
for _ in range(3):
    run()
torch.cuda.synchronize()
ev[0].record()
for _ in range(reps):
    run()
ev[1].record()
torch.cuda.synchronize()
ms = ev[0].elapsed_time(ev[1]) / reps
by = N * H * W * cp * 2 + N * Ho * Wo * cout * 2
fl = 2 * N * Ho * Wo * cout * cin * ks * ks
print(f"{mode} {cin}->{cout} k{ks} s{s} @{H}x{W} n={N}: {ms*1e3:.1f} us  {by/ms/1e6:.0f} GB/s  {fl/ms/1e9:.1f} TFLOP/s  (v1={'DY_CONV_V1' in os.environ})")
if os.environ.get("DY_TIMING"):
    import ctypes as C
    from ultralytics.hip import lib
    L = lib()
    out = (C.c_ulonglong * 8)()
    L.dy_conv_timing_fetch(out, 1)
    run(); torch.cuda.synchronize()
    L.dy_conv_timing_fetch(out, 1)
    n = max(out[7], 1)
    names = ["loop-top", "k_loop", "epilogue N-tiles", "stage_write(+vmcnt)", "prefetch_issue", "barrier", "epilogue setup (+slot idle)"]
    tot = sum(out[i] for i in range(7))
    for i, nm in enumerate(names):
        print(f"  {nm:22s} {out[i]/n:10.0f} cycles/wave  {100*out[i]/tot:5.1f}%")
    print(f"  total {tot/n:.0f} cycles per wave0, {n} workgroups")
