"""North-star clause 'mAP50 on a held-out synthetic set within +-0.2 of the reference': the protocol of
tests/golden/make_golden.py::gen_map (reference model / loss / optimizer / EMA / NMS / metrics on CPU, 30 epochs x 8 batches
of 16 planted-rectangle images, 320x320, 4 classes, shared reference-like initial state) repeated with this package's
DetectionTrainer + DetectionValidator on the GPU.  fp16 activations and a different summation order make the two runs
diverge numerically from the first step on; what must agree is what they learn."""
import os

import numpy as np
import pytest
import torch

from conftest import CFG_DIR
from golden.cases import planted_batches
from oracle import graph as og

pytestmark = pytest.mark.gpu


def test_map50_within_tolerance_of_reference(golden):
    from ultralytics.engine.trainer import DetectionTrainer
    from ultralytics.models.yolo.detect import DetectionValidator
    from ultralytics.nn.tasks import DetectionModel
    G = golden("map_parity")
    nc, imgsz, B, nb, epochs, nval, seed = [int(v) for v in G["protocol"]]
    cfg = os.path.join(CFG_DIR, "yolov8n-ASF-P2P2.yaml")
    y = og.load_yaml(cfg)
    y["nc"] = nc
    m = DetectionModel(cfg, ch=3, nc=nc, verbose=False)
    m.load_state_dict(og.default_init_state(og.build_graph(y), seed=seed), strict=True)
    train = [{k: torch.from_numpy(v) for k, v in b.items()} for b in planted_batches(1, nb, B, imgsz, nc)]
    val = [{k: torch.from_numpy(v) for k, v in b.items()} for b in planted_batches(2, nval, B, imgsz, nc)]
    tr = DetectionTrainer(m, overrides=dict(batch=B, imgsz=imgsz, epochs=epochs, hipgraph=False, optimizer="SGD"))  # gen_map drives SGD
    hist = np.stack([np.asarray(h, dtype=np.float64) for h in tr.train(train, B, imgsz, log_every=1)])
    ref_hist = G["loss_hist"]
    for e in range(epochs):
        print(f"  epoch {e + 1:2d}  ours {hist[e].round(3)}  reference {ref_hist[e].round(3)}")
    print("loss items, epoch 1 / 10 / 30   ours:", hist[0].round(3), hist[9].round(3), hist[-1].round(3))
    print("                           reference:", ref_hist[0].round(3), ref_hist[9].round(3), ref_hist[-1].round(3))
    # same protocol, same start: the first epoch's mean losses agree closely, the last within 15 %
    assert np.abs(hist[0] - ref_hist[0]).max() / ref_hist[0].max() < 0.05
    assert np.abs(hist[-1] - ref_hist[-1]).max() / ref_hist[-1].max() < 0.15
    res = DetectionValidator(dataloader=val, args=dict(conf=0.001, iou=0.7))(model=tr.ema.ema)
    p, r, m50, m5095 = G["mean_results"]
    print(f"held-out mAP50 ours {res['metrics/mAP50(B)']:.4f} vs reference {m50:.4f}; mAP50-95 {res['metrics/mAP50-95(B)']:.4f} vs {m5095:.4f}; "
          f"P {res['metrics/precision(B)']:.3f} vs {p:.3f}; R {res['metrics/recall(B)']:.3f} vs {r:.3f}")
    assert abs(res["metrics/mAP50(B)"] - m50) < 0.2          # the north-star's bound
    # What is actually observed.  The protocol is one 240-step trajectory of a chaotic system: 24 repetitions of this run that differ
    # only in summation order / loss-scale start (DY_WGRAD_TPB = 1..24, loss_scale = 2^6..2^18; round 2, scratch measurement) gave
    # mAP50 0.670 +- 0.031 (0.609 .. 0.741) and mAP50-95 0.514 +- 0.032 (0.441 .. 0.578) against the reference's single run
    # (0.694 / 0.557).  The tree's default settings are deterministic and land at 0.680 / 0.505; the bounds cover the spread.
    assert abs(res["metrics/mAP50(B)"] - m50) < 0.12 and abs(res["metrics/mAP50-95(B)"] - m5095) < 0.13
