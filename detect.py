#!/usr/bin/env python3
"""Entry script with the reference's shape (reference detect.py:5-14): load weights or a YAML, predict, soft-NMS.

    python detect.py [weights.pt | model.yaml] [image ...]      (no image: four random 640x640 tensors)
"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "experiment-yolo_amd"))
from ultralytics import YOLO  # noqa: E402

if __name__ == "__main__":
    model = YOLO(sys.argv[1] if len(sys.argv) > 1 else "yolov8n-ASF-P2P2.yaml")
    if len(sys.argv) > 2:
        import numpy as np
        from PIL import Image
        from ultralytics.data.dataset import letterbox_geometry
        for path in sys.argv[2:]:
            img = np.asarray(Image.open(path).convert("RGB"))
            H, W = img.shape[:2]
            r, new_unpad, _, (top, bottom, left, right) = letterbox_geometry((H, W), (640, 640), scaleup=True)
            x = torch.from_numpy(img).permute(2, 0, 1)[None].float()
            x = torch.nn.functional.interpolate(x, size=(new_unpad[1], new_unpad[0]), mode="bilinear", align_corners=False)
            x = torch.nn.functional.pad(x, (left, right, top, bottom), value=114.0) / 255
            det = model.predict(x, imgsz=640, conf=0.25, iou=0.7)[0]
            det[:, [0, 2]] = ((det[:, [0, 2]] - left) / r).clamp(0, W)  # back to the image (ops.scale_boxes)
            det[:, [1, 3]] = ((det[:, [1, 3]] - top) / r).clamp(0, H)
            print(path, tuple(det.shape), det[:5].tolist())
    else:
        out = model.predict(torch.rand(4, 3, 640, 640), imgsz=640, conf=0.25, iou=0.7)
        print([tuple(o.shape) for o in out])
