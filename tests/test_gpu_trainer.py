"""-m gpu: the training loop (warm-up, accumulate, all-reduce hook, optimizer + EMA, hipGraph replay) learns on a synthetic
source, the EMA export and the predict / NMS facade run, and the vanilla P2 model of BASELINE.json configs[4] runs fused."""
import os

import pytest
import torch

from conftest import CFG_DIR

pytestmark = pytest.mark.gpu


def test_trainer_learns_and_exports():
    from ultralytics import YOLO
    from ultralytics.data import SyntheticDetection
    torch.manual_seed(0)
    y = YOLO(os.path.join(CFG_DIR, "yolov8n-ASF-P2P2.yaml"))
    src = SyntheticDetection(n_batches=12, batch=4, imgsz=64, boxes_per_image=3, wh=(0.1, 0.4), seed=3)
    hist = y.train(data=src, batch=4, imgsz=64, epochs=3, optimizer="SGD", warmup_epochs=0.0, lr0=0.01, nbs=4, hipgraph=True)
    first, last = float(sum(hist[0])), float(sum(hist[-1]))
    assert all(torch.isfinite(h).all() for h in hist)
    assert last < first, (first, last)  # it learns something in 36 steps
    plan = y.trainer.plan
    assert float(plan.state[5]) >= 30  # optimizer steps actually taken (a few may be skipped while the loss scale settles)
    ema = y.trainer.ema.ema
    w = dict(y.model.named_parameters())["model.0.conv.weight"]
    we = dict(ema.named_parameters())["model.0.conv.weight"]
    assert we.shape == w.shape and not torch.equal(we.to(w.device), w)
    out = y.predict(torch.rand(2, 3, 64, 64), conf=0.001)
    assert len(out) == 2 and all(o.shape[1] == 6 for o in out)


def test_p2_model_fused_inference():
    from ultralytics.nn.tasks import DetectionModel
    m = DetectionModel(os.path.join(CFG_DIR, "yolov8n-p2.yaml"), verbose=False).cuda().eval()
    m.fuse()
    with torch.no_grad():
        y, feats = m(torch.rand(2, 3, 128, 128).cuda())
    A = 32 * 32 + 16 * 16 + 8 * 8 + 4 * 4
    assert y.shape == (2, 84, A) and len(feats) == 4 and torch.isfinite(y).all()
