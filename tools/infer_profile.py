#!/usr/bin/env python3
"""Per-launch device-time table of one recorded eval forward (InferPlan): infer_profile.py <model yaml> <batch> <imgsz> [fuse]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "experiment-yolo_amd")]
import torch  # noqa: E402

from ultralytics import YOLO  # noqa: E402
from ultralytics.hip.infer import InferPlan  # noqa: E402

name, B, S = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
torch.manual_seed(0)
m = YOLO(name).model.cuda().eval()
if len(sys.argv) > 4 and sys.argv[4] == "fuse":
    m.fuse()
plan = InferPlan(m, B, S, S, use_graph=False)
x = torch.rand(B, 3, S, S, device="cuda")
for _ in range(3):
    plan(x)
ops = [o for o in plan.rec.ops if o[0] is not None]
s = plan.eng.stream
reps = 5
evs = [torch.cuda.Event(enable_timing=True) for _ in range(len(ops) + 2)]
tot = [0.0] * (len(ops) + 1)
for _ in range(reps):
    torch.cuda._sleep(60_000_000)
    evs[0].record()
    for i, (fn, args, nm, _sid) in enumerate(ops):
        fn(*args, s)
        evs[i + 1].record()
    plan._finish()
    evs[len(ops) + 1].record()
    torch.cuda.synchronize()
    for i in range(len(ops) + 1):
        tot[i] += evs[i].elapsed_time(evs[i + 1])
rows = []
for i, (fn, a, nm, _sid) in enumerate(ops):
    d, by = "", 0
    if nm == "dy_conv_forward":
        n, h, w, cin, cout, ks, st, dil = a[7:15]
        ho, wo = (h + 2 * (ks // 2) - ks) // st + 1, (w + 2 * (ks // 2) - ks) // st + 1
        by = n * h * w * cin * 2 + n * ho * wo * cout * (4 if a[17] & 8 else 2)
        d = f"{cin}->{cout} k{ks} s{st} @{h}x{w} epi{a[17]}  {2.0 * n * ho * wo * cin * cout * ks * ks / (tot[i] / reps) / 1e9:5.0f}TF"
    rows.append((tot[i] / reps, nm, d, by))
rows.append((tot[len(ops)] / reps, "<tail: dy_head_infer_levels / decode>", "", 0))
print(f"total device ms/forward {sum(r[0] for r in rows):.3f} ({len(rows)} launches), {B / sum(r[0] for r in rows) * 1e3:.0f} FPS device-bound")
for i, (ms, nm, d, by) in enumerate(rows):
    print(f"{i:4d} {ms * 1e3:9.1f} us  {nm:28s} {d:44s} {by / ms / 1e6 if by else 0:8.0f} GB/s")
