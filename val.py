#!/usr/bin/env python3
"""Entry script with the reference's shape (reference val.py:5-16, which validates an RT-DETR checkpoint -- another model family;
here: the detect task this package covers): load weights or a model YAML, validate on a split of a YOLO-format dataset.

    python val.py <weights.pt | model.yaml> <data.yaml> [split=val] [batch=16]
"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "experiment-yolo_amd"))
from ultralytics import YOLO  # noqa: E402

if __name__ == "__main__":
    if len(sys.argv) < 3:
        raise SystemExit(__doc__)
    model = YOLO(sys.argv[1])
    metrics = model.val(data=sys.argv[2], split=sys.argv[3] if len(sys.argv) > 3 else "val", imgsz=640,
                        batch=int(sys.argv[4]) if len(sys.argv) > 4 else 16)
    print({k: round(float(v), 4) for k, v in metrics.items()})
