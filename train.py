#!/usr/bin/env python3
"""Entry script with the reference's shape (reference train.py:6-25): build from a model YAML, train.

    python train.py path/to/data.yaml      YOLO-format dataset (images/ + labels/ folders), the decoded set kept in HBM, the default
                                           augmentation set (mosaic, affine, HSV, flips) composed on the device
    python train.py                        the synthetic source of SURVEY.md section 8(d)
"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "experiment-yolo_amd"))
from ultralytics import YOLO  # noqa: E402
from ultralytics.data import SyntheticDetection  # noqa: E402

if __name__ == "__main__":
    model = YOLO("yolov8n-ASF-P2P2.yaml")
    data = sys.argv[1] if len(sys.argv) > 1 else SyntheticDetection(n_batches=20, batch=64, imgsz=640, device="cuda:0")
    extra = dict(cache="hbm") if isinstance(data, str) else {}
    model.train(data=data, imgsz=640, **extra, epochs=2, batch=64, close_mosaic=10,
                workers=8, device="0", optimizer="SGD", project="runs/train", name="exp")
