"""Shared by tests/test_gpu_dp.py and its rank processes (tests/dp_worker.py): model, global batches and schedule."""
import os

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CFG = os.path.join(ROOT, "experiment-yolo_amd", "ultralytics", "cfg", "models", "yolov8n-ASF-P2P2.yaml")
STEPS = [int(v) for v in os.environ.get("DY_TEST_DP_STEPS", "1,2,1").split(",")]  # micro-batches accumulated before each optimizer step


def build_model():
    from ultralytics.nn.tasks import DetectionModel
    torch.manual_seed(0)
    m = DetectionModel(CFG, verbose=False)
    for k, v in m.named_parameters():
        v.requires_grad = ".dfl" not in k
    return m


def global_batch(it, micro, B, n_per=3):
    rng = np.random.default_rng(1000 + 10 * it + micro)
    n = B * n_per
    return dict(img=torch.from_numpy(rng.random((B, 3, 64, 64), dtype=np.float32)), batch_idx=torch.arange(B).repeat_interleave(n_per).float(),
                cls=torch.from_numpy(rng.integers(0, 6, (n, 1)).astype(np.float32)),
                bboxes=torch.from_numpy(np.concatenate([rng.random((n, 2)) * 0.6 + 0.2, rng.random((n, 2)) * 0.3 + 0.05], 1).astype(np.float32)))


def hyper(it):
    return [0.05, 0.01, 0.01], 0.9, [0.0, 5e-4, 0.0]
