import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "experiment-yolo_amd")]
import torch
from ultralytics import YOLO
from ultralytics.data import SyntheticDetection
for name, imgsz, batch in [("yolov8n-ASF-P2P2.yaml", 640, 16), ("yolov8n-ASF-P2.yaml", 512, 8), ("yolov8n-LD-P2.yaml", 384, 8)]:
    model = YOLO(name)
    r = model.train(data=SyntheticDetection(n_batches=12, batch=batch, imgsz=imgsz), imgsz=imgsz, epochs=3, batch=batch, device="0",
                    optimizer="SGD", project="/tmp/runs", name="s")
    print(name, imgsz, batch, "->", r if not isinstance(r, dict) else {k: (round(v, 4) if isinstance(v, float) else v) for k, v in r.items()})
