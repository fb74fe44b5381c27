#!/usr/bin/env python3
"""Entry script with the reference's shape (reference detect.py:5-14): load weights (or a model YAML), predict on a folder.

    python detect.py [weights.pt | model.yaml] [image | directory | glob]      (no source: four random 640x640 tensors)
"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "experiment-yolo_amd"))
from ultralytics import YOLO  # noqa: E402

if __name__ == "__main__":
    model = YOLO(sys.argv[1] if len(sys.argv) > 1 else "yolov8n-ASF-P2P2.yaml")  # select your model.pt path
    source = sys.argv[2] if len(sys.argv) > 2 else torch.rand(4, 3, 640, 640)
    results = model.predict(source=source,
                            imgsz=640,
                            project="runs/detect",
                            name="exp",
                            verbose=True,
                            # conf=0.2,
                            )
    for r in results[:8]:
        print(r.path, r.orig_shape, tuple(r.boxes.data.shape), r.boxes.data[:3].tolist())
