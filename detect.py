#!/usr/bin/env python3
"""Entry script with the reference's shape (reference detect.py:5-14): load weights or a YAML, predict, soft-NMS."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "experiment-yolo_amd"))
from ultralytics import YOLO  # noqa: E402

if __name__ == "__main__":
    model = YOLO(sys.argv[1] if len(sys.argv) > 1 else "yolov8n-ASF-P2P2.yaml")
    out = model.predict(torch.rand(4, 3, 640, 640), imgsz=640, conf=0.25, iou=0.7)
    print([tuple(o.shape) for o in out])
