#!/usr/bin/env python3
"""(Lives under tests/ because it times the CPU oracle beside the HIP path: only tests/, smoke() and bench.py's cpu_baseline
leg may touch oracle/.)  Validation-path measurement: images/s of DetectionValidator.update_metrics (one
dy_match_predictions launch per batch) next to the CPU oracle's per-image loop (the reference's structure: IoU matrix to the
host, numpy sort/unique per threshold), and an end-to-end YOLO.val() on synthetic batches.
usage: val_bench.py [batch] [dets_per_image] [labels_per_image]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "experiment-yolo_amd"), os.path.join(ROOT, "tests", "golden")]
import numpy as np  # noqa: E402
import torch  # noqa: E402

from oracle import metrics as om  # noqa: E402  (checker / CPU baseline only)
from ultralytics.models.yolo.detect import DetectionValidator  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
ND = int(sys.argv[2]) if len(sys.argv) > 2 else 300
NL = int(sys.argv[3]) if len(sys.argv) > 3 else 40
rng = np.random.default_rng(0)
lab = np.concatenate([rng.random((B * NL, 2)) * 0.8 + 0.1, rng.random((B * NL, 2)) * 0.15 + 0.02], 1).astype(np.float32)
bidx = np.repeat(np.arange(B), NL).astype(np.float32)
cls = rng.integers(0, 6, (B * NL, 1)).astype(np.float32)
preds = []
for i in range(B):
    src = rng.integers(0, NL, ND)
    bx = om.xywhn_to_xyxy(lab[i * NL + src], 640, 640) + rng.normal(0, 6, (ND, 4)).astype(np.float32)
    conf = np.sort(rng.random((ND, 1)).astype(np.float32), 0)[::-1]
    preds.append(np.concatenate([bx, conf, cls[i * NL + src]], 1).astype(np.float32))
batch = dict(batch_idx=bidx, cls=cls, bboxes=lab)
tb = {k: torch.from_numpy(v).cuda() for k, v in batch.items()}
tb["img"] = torch.zeros(B, 3, 640, 640, device="cuda")
tp_list = [torch.from_numpy(p).cuda() for p in preds]
v = DetectionValidator(args=None)
v.device, v.nc, v.names = torch.device("cuda:0"), 6, {i: str(i) for i in range(6)}
v.metrics.names = v.names
for _ in range(3):
    tp = v.update_metrics(tp_list, tb)
torch.cuda.synchronize()
reps = 20
t0 = time.perf_counter()
for _ in range(reps):
    tp = v.update_metrics(tp_list, tb)
torch.cuda.synchronize()
gpu = (time.perf_counter() - t0) / reps
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
t0 = time.perf_counter()
st = om.validate_batch(preds, batch)
cpu = time.perf_counter() - t0
ref = np.concatenate(st["tp"], 0)
ok = (tp.cpu().numpy().astype(bool) == ref).all()
print(f"update_metrics  B={B} dets/img={ND} labels/img={NL}: HIP {gpu*1e3:.2f} ms/batch ({B/gpu:,.0f} img/s, incl. host packing)  |  "
      f"CPU oracle loop {cpu*1e3:.1f} ms/batch ({B/cpu:,.0f} img/s)  |  identical tp: {ok}")
v.stats = {k: val[-1:] for k, val in v.stats.items()}
t0 = time.perf_counter()
res = v.get_stats()
print(f"get_stats (host ap_per_class over {B*ND} detections): {(time.perf_counter()-t0)*1e3:.1f} ms -> mAP50 {res['metrics/mAP50(B)']:.4f} mAP50-95 {res['metrics/mAP50-95(B)']:.4f}")
