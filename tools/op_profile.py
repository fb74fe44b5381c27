#!/usr/bin/env python3
"""Per-launch device-time table of one DEAL-YOLO-N training step (event-timed replay of the recorded launch list)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "experiment-yolo_amd")]
import torch  # noqa: E402

from bench import CFG, synth_batch  # noqa: E402
from ultralytics.hip.train import StepPlan  # noqa: E402
from ultralytics.nn.tasks import DetectionModel  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
cfg = os.path.join(os.path.dirname(CFG), sys.argv[2] + ".yaml") if len(sys.argv) > 2 else CFG  # e.g. op_profile.py 64 yolov8n-LD-P2
torch.manual_seed(0)
model = DetectionModel(cfg, verbose=False).cuda().train()
plan = StepPlan(model, B, 640, nmax=8)
batch = {k: v.cuda() for k, v in synth_batch(1, B, 640, 6).items()}
plan.set_hyper([0.01] * 3, 0.937, [0, 5e-4, 0])
for _ in range(3):
    plan.forward_backward(batch)
    plan.optimizer_step()
prof = plan.profile_ops(5)
rows = []
for name, a, ms in prof:
    if name == "dy_conv_forward":
        n, h, w, cin, cout, ks, s, dil = a[7:15]
        d = f"{cin}->{cout} k{ks} s{s} dil{dil} @{h}x{w} epi{a[17]}"
        by = plan.conv_algorithmic_bytes(a)
        pp = ks // 2 * dil
        fl = 2.0 * n * ((h + 2 * pp - dil * (ks - 1) - 1) // s + 1) * ((w + 2 * pp - dil * (ks - 1) - 1) // s + 1) * cin * cout * ks * ks
        d += f" {fl / ms / 1e9:6.0f}TF"
    elif name == "dy_conv_wgrad":
        n, h, w, cin, cout, ks, s = a[6:13]
        d = f"{cin}->{cout} k{ks} s{s} @{h}x{w}"
        p = ks // 2
        by = n * h * w * cin * 2 + n * ((h + 2 * p - ks) // s + 1) * ((w + 2 * p - ks) // s + 1) * cout * 2
    elif name == "dy_conv_wgrad_bn":  # reads x, dy, raw; writes d(raw)
        n, h, w, cin, cout, ks, s = a[14:21]
        d = f"{cin}->{cout} k{ks} s{s} @{h}x{w}"
        p = ks // 2
        by = n * h * w * cin * 2 + 3 * n * ((h + 2 * p - ks) // s + 1) * ((w + 2 * p - ks) // s + 1) * cout * 2
    elif name == "dy_ldconv_sample_backward":
        d = f"C={a[15]} Np={a[16]} s{a[17]} {a[11]}x{a[12]}->{a[13]}x{a[14]} n={a[10]} dx={'y' if a[7] else 'n'}"
        by = a[10] * a[13] * a[14] * a[16] * a[15] * 2
    elif name == "dy_ldconv_sample_backward_gather":
        d = f"C={a[20]} Np={a[21]} s{a[22]} {a[16]}x{a[17]}->{a[18]}x{a[19]} n={a[15]}"
        by = a[15] * (a[18] * a[19] * a[21] + 2 * a[16] * a[17]) * a[20] * 2  # dxo + x (offset gradient) in, dx out
    elif name in ("dy_bn_act_apply",):
        d = f"npix={a[7]} C={a[8]}"
        by = a[7] * a[8] * 2 * 2
    elif name == "dy_bn_act_apply_acc":
        d = f"npix={a[12]} C={a[13]}"
        by = a[12] * a[13] * 2 * 2
    elif name == "dy_bn_act_bwd_apply":
        d = f"npix={a[8]} C={a[9]}"
        by = a[8] * a[9] * 2 * 3
    elif name == "dy_bn_act_bwd_apply_acc":
        d = f"npix={a[10]} C={a[11]}"
        by = a[10] * a[11] * 2 * 3
    elif name == "dy_bn_act_bwd_reduce":
        d = f"npix={a[7]} C={a[8]}"
        by = a[7] * a[8] * 2 * 2
    elif name == "dy_bn_act_bwd_reduce_acc":
        d = f"npix={a[6]} C={a[7]}"
        by = a[6] * a[7] * 2 * 2
    else:
        d, by = "", 0
    rows.append((ms, name, d, by))
tot = sum(r[0] for r in rows)
print(f"total device ms/step {tot:.2f}  ({len(rows)} calls)")
for ms, name, d, by in sorted(rows, key=lambda r: -r[0])[:int(os.environ.get('TOP', '60'))]:
    print(f"{ms*1e3:9.1f} us  {name:24s} {d:48s} {by/ms/1e6 if by else 0:8.0f} GB/s")
if os.environ.get("ORDER"):  # the same table in EXECUTION order (forward, loss, backward), every launch
    print("---- execution order")
    for i, (ms, name, d, by) in enumerate(rows):
        print(f"{i:4d} {ms*1e3:9.1f} us  {name:24s} {d:48s} {by/ms/1e6 if by else 0:8.0f} GB/s")
