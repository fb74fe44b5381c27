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


def box_iou(box1, box2, eps=1e-7):
    """(N,4) x (M,4) xyxy -> (N,M) IoU (reference utils/metrics.py:53-73); CUDA tensors only."""
    if box1.device.type != "cuda" or box2.device.type != "cuda":
        raise RuntimeError("box_iou: HIP path only (no CPU fallback)")
    if eps != 1e-7:
        raise NotImplementedError("box_iou: eps is fixed at the reference default 1e-7")
    a, b = box1.float().contiguous(), box2.float().contiguous()
    out = torch.empty((a.shape[0], b.shape[0]), dtype=torch.float32, device=a.device)
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
    mrec = np.concatenate(([0.0], recall, [1.0]))
    mpre = np.concatenate(([1.0], precision, [0.0]))
    mpre = np.flip(np.maximum.accumulate(np.flip(mpre)))
    grid = np.linspace(0, 1, 101)
    integrate = getattr(np, "trapezoid", None) or np.trapz
    return integrate(np.interp(grid, mrec, mpre), grid), mpre, mrec


def ap_per_class(tp, conf, pred_cls, target_cls, plot=False, on_plot=None, save_dir=None, names=(), eps=1e-16, prefix=""):
    """Per-class AP and the max-F1 operating point (reference :1142-1230); same 12-tuple, plotting not available."""
    if plot:
        raise NotImplementedError("PR-curve plotting is control plane (SURVEY.md section 8: out of scope)")
    order = np.argsort(-conf)
    tp, conf, pred_cls = tp[order], conf[order], pred_cls[order]
    unique_classes, nt = np.unique(target_cls, return_counts=True)
    nc = unique_classes.shape[0]
    x, prec_values = np.linspace(0, 1, 1000), []
    ap, p_curve, r_curve = np.zeros((nc, tp.shape[1])), np.zeros((nc, 1000)), np.zeros((nc, 1000))
    for ci, c in enumerate(unique_classes):
        sel = pred_cls == c
        n_l, n_p = nt[ci], sel.sum()
        if n_p == 0 or n_l == 0:
            continue
        fpc, tpc = (1 - tp[sel]).cumsum(0), tp[sel].cumsum(0)
        recall = tpc / (n_l + eps)
        r_curve[ci] = np.interp(-x, -conf[sel], recall[:, 0], left=0)  # negated: xp must increase
        precision = tpc / (tpc + fpc)
        p_curve[ci] = np.interp(-x, -conf[sel], precision[:, 0], left=1)
        for j in range(tp.shape[1]):
            ap[ci, j] = compute_ap(recall[:, j], precision[:, j])[0]
    f1_curve = 2 * p_curve * r_curve / (p_curve + r_curve + eps)
    k = smooth(f1_curve.mean(0), 0.1).argmax()
    p, r, f1 = p_curve[:, k], r_curve[:, k], f1_curve[:, k]
    tpn = (r * nt).round()
    fpn = (tpn / (p + eps) - tpn).round()
    return tpn, fpn, p, r, f1, ap, unique_classes.astype(int), p_curve, r_curve, f1_curve, x, np.array(prec_values)


class Metric:
    """Per-class results container (reference :1233-1402)."""

    def __init__(self):
        self.p, self.r, self.f1, self.all_ap, self.ap_class_index, self.nc = [], [], [], [], [], 0

    @property
    def ap50(self):
        return self.all_ap[:, 0] if len(self.all_ap) else []

    @property
    def ap(self):
        return self.all_ap.mean(1) if len(self.all_ap) else []

    @property
    def mp(self):
        return self.p.mean() if len(self.p) else 0.0

    @property
    def mr(self):
        return self.r.mean() if len(self.r) else 0.0

    @property
    def map50(self):
        return self.all_ap[:, 0].mean() if len(self.all_ap) else 0.0

    @property
    def map75(self):
        return self.all_ap[:, 5].mean() if len(self.all_ap) else 0.0

    @property
    def map(self):
        return self.all_ap.mean() if len(self.all_ap) else 0.0

    def mean_results(self):
        return [self.mp, self.mr, self.map50, self.map]

    def class_result(self, i):
        return self.p[i], self.r[i], self.ap50[i], self.ap[i]

    @property
    def maps(self):
        maps = np.zeros(self.nc) + self.map
        for i, c in enumerate(self.ap_class_index):
            maps[c] = self.ap[i]
        return maps

    def fitness(self):
        return (np.array(self.mean_results()) * [0.0, 0.0, 0.1, 0.9]).sum()

    def update(self, results):
        self.p, self.r, self.f1, self.all_ap, self.ap_class_index = results[:5]


class DetMetrics:
    """Reference :1405-1480: ``process`` the concatenated statistics, then read ``results_dict`` / ``mean_results``."""

    def __init__(self, save_dir=None, plot=False, on_plot=None, names=()):
        self.save_dir, self.plot, self.on_plot, self.names = save_dir, plot, on_plot, names
        self.box = Metric()
        self.speed = {"preprocess": 0.0, "inference": 0.0, "loss": 0.0, "postprocess": 0.0}
        self.task = "detect"

    def process(self, tp, conf, pred_cls, target_cls):
        results = ap_per_class(tp, conf, pred_cls, target_cls, plot=False, names=self.names)[2:]
        self.box.nc = len(self.names)
        self.box.update(results)

    @property
    def keys(self):
        return ["metrics/precision(B)", "metrics/recall(B)", "metrics/mAP50(B)", "metrics/mAP50-95(B)"]

    def mean_results(self):
        return self.box.mean_results()

    def class_result(self, i):
        return self.box.class_result(i)

    @property
    def maps(self):
        return self.box.maps

    @property
    def fitness(self):
        return self.box.fitness()

    @property
    def ap_class_index(self):
        return self.box.ap_class_index

    @property
    def results_dict(self):
        return dict(zip(self.keys + ["fitness"], self.mean_results() + [self.fitness]))
