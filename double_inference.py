#!/usr/bin/env python3
"""Entry script with the reference's shape (reference double_inference.py:494-560 without its Kaggle paths, JSON dumps, metric
tables and drawings): first-stage detections of one image, then the two-stage refinement + per-class NMS on the GPU.

    python double_inference.py <weights.pt | model.yaml> <image> [conf=0.25]
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "experiment-yolo_amd"))
from ultralytics import YOLO  # noqa: E402
from ultralytics.data.dataset import letterbox_geometry  # noqa: E402
from ultralytics.utils.double_inference import CONF_THRESHOLD, double_inference  # noqa: E402

if __name__ == "__main__":
    from PIL import Image
    model = YOLO(sys.argv[1] if len(sys.argv) > 1 else "yolov8n-ASF-P2P2.yaml")
    conf = float(sys.argv[3]) if len(sys.argv) > 3 else CONF_THRESHOLD
    img = np.asarray(Image.open(sys.argv[2]).convert("RGB")) if len(sys.argv) > 2 else np.random.default_rng(0).integers(0, 256, (720, 1280, 3), dtype=np.uint8)
    H, W = img.shape[:2]
    # first stage on the 640 letterbox of the image (detect/predict.py pre/post-processing), boxes mapped back to the image
    r, new_unpad, (dw, dh), (top, bottom, left, right) = letterbox_geometry((H, W), (640, 640), scaleup=True)
    x = torch.from_numpy(img).permute(2, 0, 1)[None].float()
    x = torch.nn.functional.interpolate(x, size=(new_unpad[1], new_unpad[0]), mode="bilinear", align_corners=False)
    x = torch.nn.functional.pad(x, (left, right, top, bottom), value=114.0) / 255
    det = model.predict(x, conf=conf, iou=0.7)[0].cpu().numpy()
    det[:, [0, 2]] = ((det[:, [0, 2]] - left) / r).clip(0, W)
    det[:, [1, 3]] = ((det[:, [1, 3]] - top) / r).clip(0, H)
    single = {"boxes": det[:, :4].tolist(), "scores": det[:, 4].tolist(), "labels": det[:, 5].astype(int).tolist()}
    refined, dt = double_inference(torch.from_numpy(img), model.model, single, conf_threshold=conf)
    print(f"single stage: {len(single['boxes'])} detections; double stage: {len(refined['boxes'])} after refinement + NMS "
          f"({dt * 1e3:.1f} ms extra)")
