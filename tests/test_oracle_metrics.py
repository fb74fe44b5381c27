"""Validation-path oracle (oracle/metrics.py) against fixtures produced by the reference's own box_iou / match_predictions /
ap_per_class / DetMetrics (tests/golden/make_golden.py::gen_metrics), and the product's host-side mirror of the same
arithmetic (ultralytics.utils.metrics) against both.  CPU only."""
import numpy as np
import pytest

from golden.cases import metric_cases, metric_geometry, synth_detections
from oracle import metrics as om


@pytest.mark.parametrize("case", metric_cases(), ids=lambda c: c[0])
def test_oracle_metrics_vs_reference(golden, case):
    name, seed, n_images, nc, ml, md, jit = case
    G = golden("metrics")
    batch, preds = synth_detections(seed, n_images, nc, ml, md, jit)
    geo = metric_geometry(name, n_images)
    for si, pred in enumerate(preds):
        idx = batch["batch_idx"] == si
        if len(pred) and idx.sum():
            tbox = om.scale_boxes((640, 640), om.xywhn_to_xyxy(batch["bboxes"][idx], 640, 640), *geo[si])
            pb = om.scale_boxes((640, 640), pred[:, :4], *geo[si])
            iou = om.box_iou(tbox, pb)
            assert np.abs(iou - G[f"{name}/iou{si}"]).max() < 1e-6
            tp = om.match_predictions(pred[:, 5], batch["cls"].reshape(-1)[idx], iou)
            assert (tp == G[f"{name}/tp{si}"]).all()
    st = {k: np.concatenate(v, 0) for k, v in om.validate_batch(preds, batch, geometry=geo).items()}
    res = om.ap_per_class(st["tp"], st["conf"], st["pred_cls"], st["target_cls"])
    for k, gk in (("ap", "ap"), ("p", "p"), ("r", "r"), ("f1", "f1"), ("tp", "tp_c"), ("fp", "fp_c"), ("p_curve", "p_curve"),
                  ("r_curve", "r_curve"), ("f1_curve", "f1_curve")):
        assert np.allclose(res[k], G[f"{name}/{gk}"], atol=1e-12), k
    assert (res["classes"] == G[f"{name}/classes"]).all()
    assert np.allclose(om.mean_results(res), G[f"{name}/mean_results"], atol=1e-12)
    assert abs(om.fitness(res) - float(G[f"{name}/fitness"])) < 1e-12


@pytest.mark.parametrize("case", metric_cases(), ids=lambda c: c[0])
def test_host_ap_per_class_mirror_vs_reference(golden, case):
    """ultralytics.utils.metrics.ap_per_class / DetMetrics (host arithmetic of the product) on the oracle's statistics."""
    from ultralytics.utils.metrics import DetMetrics, ap_per_class
    name, seed, n_images, nc, ml, md, jit = case
    G = golden("metrics")
    batch, preds = synth_detections(seed, n_images, nc, ml, md, jit)
    st = {k: np.concatenate(v, 0) for k, v in om.validate_batch(preds, batch, geometry=metric_geometry(name, n_images)).items()}
    out = ap_per_class(st["tp"], st["conf"], st["pred_cls"], st["target_cls"])
    for got, gk in zip(out[:10], ("tp_c", "fp_c", "p", "r", "f1", "ap", "classes", "p_curve", "r_curve", "f1_curve")):
        assert np.allclose(got, G[f"{name}/{gk}"], atol=1e-12), gk
    dm = DetMetrics(names={i: str(i) for i in range(nc)})
    dm.process(st["tp"], st["conf"], st["pred_cls"], st["target_cls"])
    assert np.allclose(dm.mean_results(), G[f"{name}/mean_results"], atol=1e-12)
    assert abs(dm.fitness - float(G[f"{name}/fitness"])) < 1e-12
