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
from ultralytics.utils.double_inference import CONF_THRESHOLD, double_inference  # noqa: E402

if __name__ == "__main__":
    from PIL import Image
    model = YOLO(sys.argv[1] if len(sys.argv) > 1 else "yolov8n-ASF-P2P2.yaml")
    conf = float(sys.argv[3]) if len(sys.argv) > 3 else CONF_THRESHOLD
    img = np.asarray(Image.open(sys.argv[2]).convert("RGB")) if len(sys.argv) > 2 else np.random.default_rng(0).integers(0, 256, (720, 1280, 3), dtype=np.uint8)
    # first stage: model.predict on the image (letterbox, forward, soft-NMS, boxes mapped back: detect/predict.py:23-43)
    det = model.predict(source=np.ascontiguousarray(img[..., ::-1]), imgsz=640, conf=conf, iou=0.7)[0].boxes.data.cpu().numpy()
    single = {"boxes": det[:, :4].tolist(), "scores": det[:, 4].tolist(), "labels": det[:, 5].astype(int).tolist()}
    refined, dt = double_inference(torch.from_numpy(img), model.model, single, conf_threshold=conf)
    print(f"single stage: {len(single['boxes'])} detections; double stage: {len(refined['boxes'])} after refinement + NMS "
          f"({dt * 1e3:.1f} ms extra)")
