"""Detection metrics with the reference's names and return conventions (reference ultralytics/utils/metrics.py).

``box_iou`` runs on the GPU (libdealyolo_hip: dy_box_iou); the per-batch matching lives in
``ultralytics.models.yolo.detect.DetectionValidator`` (dy_match_predictions).  ``ap_per_class`` / ``compute_ap`` / ``smooth``
and the ``Metric`` / ``DetMetrics`` containers are host arithmetic in the reference as well (numpy on the concatenated
statistics, utils/metrics.py:1051-1480) and stay host arithmetic here; plotting and the confusion matrix are control plane.
"""
import ctypes as C

import numpy as np
import torch

from ..hip import check, lib
from ..hip.engine import dev_empty


def box_iou(box1, box2, eps=1e-7):
    """(N,4) x (M,4) xyxy -> (N,M) IoU (reference utils/metrics.py:53-73); CUDA tensors only."""
    if box1.device.type != "cuda" or box2.device.type != "cuda":
        raise RuntimeError("box_iou: HIP path only (no CPU fallback)")
    if eps != 1e-7:
        raise NotImplementedError("box_iou: eps is fixed at the reference default 1e-7")
    a, b = box1.float().contiguous(), box2.float().contiguous()
    out = dev_empty((a.shape[0], b.shape[0]), torch.float32, a.device)
    check(lib().dy_box_iou(a.data_ptr(), a.shape[0], b.data_ptr(), b.shape[0], out.data_ptr(),
                           torch.cuda.current_stream(a.device).cuda_stream), "dy_box_iou")
    return out


def smooth(y, f=0.05):
    """Box filter of fraction f (reference :1051-1056)."""
    nf = round(len(y) * f * 2) // 2 + 1  # odd number of taps
    edge = np.ones(nf // 2)
    padded = np.concatenate((edge * y[0], y, edge * y[-1]), 0)
    return np.convolve(padded, np.ones(nf) / nf, mode="valid")


def compute_ap(recall, precision):
    """AP by 101-point interpolation of the precision envelope (reference :1109-1139).  Returns (ap, mpre, mrec)."""
    mrec = np.r_[0.0, recall, 1.0]
    envelope = np.maximum.accumulate(np.r_[1.0, precision, 0.0][::-1])[::-1]  # right-to-left running maximum
    grid = np.linspace(0, 1, 101)
    integrate = getattr(np, "trapezoid", None) or np.trapz
    return integrate(np.interp(grid, mrec, envelope), grid), envelope, mrec


def ap_per_class(tp, conf, pred_cls, target_cls, plot=False, on_plot=None, save_dir=None, names=(), eps=1e-16, prefix=""):
    """Per-class AP and the max-F1 operating point (reference :1142-1230); same 12-tuple, plotting not available."""
    if plot:
        raise NotImplementedError("PR-curve plotting is control plane (SURVEY.md section 8: out of scope)")
    order = np.argsort(-conf)
    tp, conf, pred_cls = tp[order], conf[order], pred_cls[order]
    classes, n_labels = np.unique(target_cls, return_counts=True)
    n_cls, n_thr, n_grid = classes.shape[0], tp.shape[1], 1000
    x = np.linspace(0, 1, n_grid)
    ap = np.zeros((n_cls, n_thr))
    p_curve, r_curve = np.zeros((n_cls, n_grid)), np.zeros((n_cls, n_grid))
    for ci, (c, n_l) in enumerate(zip(classes, n_labels)):
        sel = pred_cls == c
        if not sel.any() or n_l == 0:
            continue
        hits = tp[sel].cumsum(0)
        misses = (1 - tp[sel]).cumsum(0)
        recall, precision = hits / (n_l + eps), hits / (hits + misses)
        r_curve[ci] = np.interp(-x, -conf[sel], recall[:, 0], left=0)  # negated: xp must increase
        p_curve[ci] = np.interp(-x, -conf[sel], precision[:, 0], left=1)
        ap[ci] = [compute_ap(recall[:, j], precision[:, j])[0] for j in range(n_thr)]
    f1_curve = 2 * p_curve * r_curve / (p_curve + r_curve + eps)
    k = smooth(f1_curve.mean(0), 0.1).argmax()
    p, r, f1 = p_curve[:, k], r_curve[:, k], f1_curve[:, k]
    tpn = (r * n_labels).round()
    fpn = (tpn / (p + eps) - tpn).round()
    return tpn, fpn, p, r, f1, ap, classes.astype(int), p_curve, r_curve, f1_curve, x, np.array([])


def _avg(a, empty=0.0):
    return a.mean() if len(a) else empty


class Metric:
    """Per-class results container with the reference's attribute names (reference :1233-1402): ``p, r, f1, all_ap
    (classes x 10 IoU thresholds), ap_class_index, nc`` and the derived ``ap50, ap, mp, mr, map50, map75, map, maps``."""

    def __init__(self):
        self.p, self.r, self.f1, self.all_ap, self.ap_class_index, self.nc = [], [], [], [], [], 0

    def update(self, results):
        self.p, self.r, self.f1, self.all_ap, self.ap_class_index = results[:5]

    def _ap_at(self, col):
        return self.all_ap[:, col] if len(self.all_ap) else []

    ap50 = property(lambda self: self._ap_at(0))
    ap = property(lambda self: self.all_ap.mean(1) if len(self.all_ap) else [])
    mp = property(lambda self: _avg(self.p))
    mr = property(lambda self: _avg(self.r))
    map50 = property(lambda self: _avg(self._ap_at(0)))
    map75 = property(lambda self: _avg(self._ap_at(5)))
    map = property(lambda self: _avg(self.all_ap) if len(self.all_ap) else 0.0)

    @property
    def maps(self):
        out = np.full(self.nc, self.map, dtype=np.float64)
        out[np.asarray(self.ap_class_index, dtype=int)] = self.ap
        return out

    def mean_results(self):
        return [self.mp, self.mr, self.map50, self.map]

    def class_result(self, i):
        return self.p[i], self.r[i], self.ap50[i], self.ap[i]

    def fitness(self):
        mp, mr, m50, m = self.mean_results()
        return 0.1 * m50 + 0.9 * m  # weights [0, 0, 0.1, 0.9] over (P, R, mAP50, mAP50-95)


class DetMetrics:
    """Reference :1405-1480: ``process`` the concatenated statistics, then read ``results_dict`` / ``mean_results``."""
    keys = ["metrics/precision(B)", "metrics/recall(B)", "metrics/mAP50(B)", "metrics/mAP50-95(B)"]
    task = "detect"

    def __init__(self, save_dir=None, plot=False, on_plot=None, names=()):
        self.save_dir, self.plot, self.on_plot, self.names = save_dir, plot, on_plot, names
        self.box = Metric()
        self.speed = dict.fromkeys(("preprocess", "inference", "loss", "postprocess"), 0.0)

    def process(self, tp, conf, pred_cls, target_cls):
        self.box.nc = len(self.names)
        self.box.update(ap_per_class(tp, conf, pred_cls, target_cls, plot=False, names=self.names)[2:])

    def __getattr__(self, name):  # mean_results / class_result / maps / ap_class_index are the box metric's
        if name in ("mean_results", "class_result", "maps", "ap_class_index"):
            return getattr(self.box, name)
        raise AttributeError(name)

    @property
    def fitness(self):
        return self.box.fitness()

    @property
    def results_dict(self):
        return dict(zip(self.keys + ["fitness"], [*self.box.mean_results(), self.fitness]))
