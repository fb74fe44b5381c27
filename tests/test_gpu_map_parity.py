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


class _Reordered:
    """The eight training batches in another seeded order every epoch -- gen_map_dist's neutral perturbation, draw for draw."""

    def __init__(self, batches, seed):
        self.b, self.rng = batches, np.random.default_rng(1000 + seed)

    def __len__(self):
        return len(self.b)

    def __iter__(self):
        return iter([self.b[j] for j in self.rng.permutation(len(self.b))])


def test_map50_distribution_matches_the_reference_distribution(golden):
    """One trajectory of a chaotic system against another says little (the single-run test above needs a 0.12 bound).  The reference
    itself, re-run under SIX neutral perturbations (the same batches in another order every epoch: make_golden.py::gen_map_dist,
    map_parity_dist.npz), spreads over mAP50 0.534 .. 0.732 (mean 0.638, sigma 0.067).  The same six runs here must give the same
    MEAN within the standard error of the difference (two-sigma), and no single run may leave the north-star's +-0.2 band around the
    reference mean."""
    from ultralytics.engine.trainer import DetectionTrainer
    from ultralytics.models.yolo.detect import DetectionValidator
    from ultralytics.nn.tasks import DetectionModel
    G, D = golden("map_parity"), golden("map_parity_dist")
    nc, imgsz, B, nb, epochs, nval, seed = [int(v) for v in G["protocol"]]
    ref = np.asarray(D["mean_results"], dtype=np.float64)  # (k, 4): P, R, mAP50, mAP50-95
    cfg = os.path.join(CFG_DIR, "yolov8n-ASF-P2P2.yaml")
    y = og.load_yaml(cfg)
    y["nc"] = nc
    init = og.default_init_state(og.build_graph(y), seed=seed)
    train = [{k: torch.from_numpy(v) for k, v in b.items()} for b in planted_batches(1, nb, B, imgsz, nc)]
    val = [{k: torch.from_numpy(v) for k, v in b.items()} for b in planted_batches(2, nval, B, imgsz, nc)]
    ours = []
    for s in [int(v) for v in D["order_seeds"]]:
        m = DetectionModel(cfg, ch=3, nc=nc, verbose=False)
        m.load_state_dict(init, strict=True)
        tr = DetectionTrainer(m, overrides=dict(batch=B, imgsz=imgsz, epochs=epochs, hipgraph=False, optimizer="SGD"))
        tr.train(_Reordered(train, s), B, imgsz)
        res = DetectionValidator(dataloader=val, args=dict(conf=0.001, iou=0.7))(model=tr.ema.ema)
        ours.append([res["metrics/precision(B)"], res["metrics/recall(B)"], res["metrics/mAP50(B)"], res["metrics/mAP50-95(B)"]])
        print(f"  order seed {s}: mAP50 ours {ours[-1][2]:.4f}  reference {ref[len(ours) - 1][2]:.4f}")
    ours = np.asarray(ours)
    k = len(ours)
    for j, name in ((2, "mAP50"), (3, "mAP50-95")):
        mo, mr, so, sr = ours[:, j].mean(), ref[:, j].mean(), ours[:, j].std(ddof=1), ref[:, j].std(ddof=1)
        se = float(np.sqrt(so ** 2 / k + sr ** 2 / k))
        print(f"{name}: ours {mo:.4f} +- {so:.4f}, reference {mr:.4f} +- {sr:.4f}; difference of means {mo - mr:+.4f}, standard error {se:.4f}")
        assert abs(mo - mr) <= 2.0 * se + 0.01, name
    assert np.abs(ours[:, 2] - ref[:, 2].mean()).max() < 0.2  # the north-star's bound, every run
