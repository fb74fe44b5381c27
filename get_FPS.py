#!/usr/bin/env python3
"""Latency / FPS protocol of the reference (reference get_FPS.py:33-87): fuse, optional half (the HIP path always stores
activations in fp16), 200 warm-up + 1000 timed forwards bracketed by synchronisation, no NMS."""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "experiment-yolo_amd"))
from ultralytics import YOLO  # noqa: E402
from ultralytics.utils.torch_utils import select_device  # noqa: E402

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--weights", default="yolov8n-p2.yaml")
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--imgs", nargs="+", type=int, default=[1280, 1280])
    ap.add_argument("--device", default="0")
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--testtime", type=int, default=1000)
    opt = ap.parse_args()
    dev = select_device(opt.device)
    model = YOLO(opt.weights).model.to(dev).eval()
    model.fuse()
    x = torch.rand(opt.batch, 3, *opt.imgs, device=dev)
    with torch.no_grad():
        for _ in range(opt.warmup):
            model(x)
        torch.cuda.synchronize()
        ts = []
        for _ in range(opt.testtime):
            t0 = time.perf_counter()
            model(x)
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
    t = np.array(ts)
    print(f"model {opt.weights} bs={opt.batch} {opt.imgs}: latency {t.mean() / opt.batch * 1e3:.4f} ms/img "
          f"(std {t.std() / opt.batch * 1e3:.4f}), FPS {opt.batch / t.mean():.1f}")
