#!/usr/bin/env python3
"""Time of the second stage for one 1920x1080 image with K first-stage detections (crop kernel + batched forward + soft-NMS +
refinement + merge).  Usage: two_stage_bench.py [K=64]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "experiment-yolo_amd"))
from ultralytics.nn.tasks import DetectionModel  # noqa: E402
from ultralytics.utils import double_inference as di  # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 64
rng = np.random.default_rng(0)
H, W = 1080, 1920
img = torch.from_numpy(rng.integers(0, 256, (H, W, 3), dtype=np.uint8)).cuda()
c = np.stack([rng.uniform(60, W - 60, K), rng.uniform(60, H - 60, K)], 1)
wh = rng.uniform(10, 100, (K, 2))
boxes = np.concatenate([c - wh / 2, c + wh / 2], 1)
pred = {"boxes": boxes.tolist(), "scores": rng.uniform(0.3, 0.8, K).tolist(), "labels": rng.integers(0, 6, K).tolist()}
torch.manual_seed(0)
model = DetectionModel("yolov8n-ASF-P2P2.yaml", verbose=False).cuda().eval()
model.fuse()
for _ in range(3):
    di.double_inference(img, model, pred)
torch.cuda.synchronize()
t0 = time.time()
n = 10
for _ in range(n):
    di.double_inference(img, model, pred)
torch.cuda.synchronize()
dt = (time.time() - t0) / n
print(f"two-stage refinement of {K} detections on a {W}x{H} image: {dt * 1e3:.1f} ms per image ({K / dt:.0f} crops/s)")
