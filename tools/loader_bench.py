#!/usr/bin/env python3
"""PCIe-inclusive training rate: DEAL-YOLO-N 640x640 bs=64 fed by ultralytics.data.HipDataLoader from a YOLO-format dataset
on local disk (uint8 NHWC pinned batches, H2D on a copy stream one batch ahead) next to the same step with the batch
resident in HBM.  Usage: loader_bench.py [n_images=512] [steps=60] [cache=ram|disk|hbm] [aug=0|1]   (aug=1 with hbm: mosaic 1.0,
degrees 5, translate 0.1, scale 0.5, shear 2, fliplr 0.5 composed on the device)"""
import os
import sys
import tempfile
import time
from types import SimpleNamespace

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "experiment-yolo_amd"))
from ultralytics.data import build_dataloader, build_yolo_dataset, check_det_dataset  # noqa: E402
from ultralytics.hip.train import StepPlan  # noqa: E402
from ultralytics.nn.tasks import DetectionModel  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
STEPS = int(sys.argv[2]) if len(sys.argv) > 2 else 60
MODE = sys.argv[3] if len(sys.argv) > 3 else "ram"
CACHE = {"ram": True, "disk": False, "hbm": "hbm"}[MODE]
AUG = len(sys.argv) > 4 and sys.argv[4] == "1"
B, S = 64, 640
root = tempfile.mkdtemp(prefix="dy_loader_bench_")
os.makedirs(os.path.join(root, "images", "train"))
os.makedirs(os.path.join(root, "labels", "train"))
rng = np.random.default_rng(0)
from PIL import Image  # noqa: E402
t0 = time.time()
for i in range(N):
    img = rng.integers(0, 256, (S, S, 3), dtype=np.uint8)
    if i < 8:
        Image.fromarray(img).save(os.path.join(root, "images", "train", f"{i:05d}.png"))
    else:  # a 1x1-free valid tiny PNG header costs time to verify; write the PNG small and the pixels as the *.npy cache
        Image.fromarray(img[:16, :16]).resize((S, S)).save(os.path.join(root, "images", "train", f"{i:05d}.png"))
    np.save(os.path.join(root, "images", "train", f"{i:05d}.npy"), img)
    k = 8
    lab = np.concatenate([rng.integers(0, 6, (k, 1)), rng.random((k, 2)) * 0.8 + 0.1, rng.random((k, 2)) * 0.08 + 0.01], 1)
    np.savetxt(os.path.join(root, "labels", "train", f"{i:05d}.txt"), lab, fmt=["%d", "%.6f", "%.6f", "%.6f", "%.6f"])
open(os.path.join(root, "data.yaml"), "w").write("path: .\ntrain: images/train\nval: images/train\nnc: 6\n")
print(f"dataset: {N} images written in {time.time() - t0:.1f} s", flush=True)

data = check_det_dataset(os.path.join(root, "data.yaml"))
dev = torch.device("cuda", 0)
cfg = SimpleNamespace(imgsz=S, cache=CACHE, fraction=1.0, rect=False, **(dict(mosaic=1.0, degrees=5.0, translate=0.1, scale=0.5, shear=2.0, fliplr=0.5) if AUG else {}))
ds = build_yolo_dataset(cfg, data["train"], B, data, mode="train", flip_on_device=True)
loader = build_dataloader(ds, B, 14, shuffle=True, device=dev, drop_last=True)

t0 = time.time()
nb = 0
for ep in range(max(1, STEPS // len(loader))):
    for batch in loader:
        nb += 1
torch.cuda.synchronize()
dt = time.time() - t0
TAG = MODE + (", device-side mosaic + affine + flips" if AUG else "")
print(f"loader alone ({TAG}): {nb * B / dt:.0f} images/s ({dt / nb * 1e3:.2f} ms/batch, 14 threads)", flush=True)

torch.manual_seed(0)
model = DetectionModel("yolov8n-ASF-P2P2.yaml", verbose=False).to(dev).train()
plan = StepPlan(model, B, S, nmax=64 if AUG else 8, optimizer="SGD", use_graph=True)  # a mosaic carries the boxes of four images


def step(batch):
    plan.set_hyper([0.01] * 3, 0.937, [0.0, 0.0005, 0.0])
    plan.forward_backward(batch)
    plan.optimizer_step()


first = next(iter(loader))
first = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in first.items()}
if MODE != "hbm":
    first.pop("flip", None)  # resident reference: the plain uint8 import
if MODE == "hbm":
    plan.forward_backward(first)  # record for the pool
for _ in range(5):
    step(first)
torch.cuda.synchronize()
t0 = time.time()
for _ in range(STEPS):
    step(first)
torch.cuda.synchronize()
res = STEPS * B / (time.time() - t0)
print(f"step with the batch resident in HBM (uint8 NHWC input): {res:.0f} images/s", flush=True)
n = 0
t0 = time.time()
while n < STEPS:
    for batch in loader:
        step({k: (v.to(dev, non_blocking=True) if torch.is_tensor(v) else v) for k, v in batch.items()})
        n += 1
        if n >= STEPS:
            break
torch.cuda.synchronize()
fed = n * B / (time.time() - t0)
print(f"step fed by the loader (decode/assemble -> pinned -> H2D on a copy stream -> import kernel): {fed:.0f} images/s "
      f"= {fed / res:.2f} of resident", flush=True)
