"""TEST INFRASTRUCTURE -- CPU restatement of the reference's validation arithmetic.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package; the product path
(experiment-yolo_amd/) never does.  Pinned by tests/golden/metrics.npz, which the REFERENCE's own functions produced
(tests/golden/make_golden.py::gen_metrics).

Follows, function by function:
  box_iou            ultralytics/utils/metrics.py:53-73
  match_predictions  ultralytics/engine/validator.py:217-257 (the numpy path, use_scipy=False)
  smooth             ultralytics/utils/metrics.py:1051-1056
  compute_ap         ultralytics/utils/metrics.py:1109-1139 (101-point interpolation)
  ap_per_class       ultralytics/utils/metrics.py:1142-1230
  mean_results/fitness  ultralytics/utils/metrics.py:1233-1402 (Metric), 1405-1480 (DetMetrics)
  xywhn_to_xyxy/scale_boxes  ultralytics/utils/ops.py (xywh2xyxy, scale_boxes, clip_boxes) as called by val.py:93-115
"""
import numpy as np

IOUV = np.linspace(0.5, 0.95, 10).astype(np.float32)  # torch.linspace(0.5, 0.95, 10), models/yolo/detect/val.py:37


def box_iou(box1, box2, eps=1e-7):
    """(N,4) x (M,4) xyxy -> (N,M) IoU, float32 arithmetic like the torch original."""
    b1, b2 = np.asarray(box1, np.float32), np.asarray(box2, np.float32)
    a1, a2 = b1[:, None, :2], b1[:, None, 2:]
    c1, c2 = b2[None, :, :2], b2[None, :, 2:]
    inter = np.clip(np.minimum(a2, c2) - np.maximum(a1, c1), 0, None).prod(2)
    return inter / ((a2 - a1).prod(2) + (c2 - c1).prod(2) - inter + np.float32(eps))


def match_predictions(pred_classes, true_classes, iou, iouv=IOUV):
    """Greedy matching: every detection keeps its best same-class label above the threshold, then every label keeps the
    FIRST (lowest-index = most confident) of the detections that chose it.  iou: (labels, detections)."""
    pred_classes, true_classes = np.asarray(pred_classes), np.asarray(true_classes)
    correct = np.zeros((pred_classes.shape[0], len(iouv)), dtype=bool)
    iou = np.asarray(iou, np.float32) * (true_classes[:, None] == pred_classes[None, :])
    for i, thr in enumerate([float(t) for t in iouv]):  # the reference compares against python floats of the fp32 thresholds
        matches = np.array(np.nonzero(iou >= thr)).T
        if matches.shape[0]:
            if matches.shape[0] > 1:
                matches = matches[iou[matches[:, 0], matches[:, 1]].argsort()[::-1]]
                matches = matches[np.unique(matches[:, 1], return_index=True)[1]]
                matches = matches[np.unique(matches[:, 0], return_index=True)[1]]
            correct[matches[:, 1].astype(int), i] = True
    return correct


def smooth(y, f=0.05):
    nf = round(len(y) * f * 2) // 2 + 1
    p = np.ones(nf // 2)
    yp = np.concatenate((p * y[0], y, p * y[-1]), 0)
    return np.convolve(yp, np.ones(nf) / nf, mode="valid")


def compute_ap(recall, precision):
    mrec = np.concatenate(([0.0], recall, [1.0]))
    mpre = np.concatenate(([1.0], precision, [0.0]))
    mpre = np.flip(np.maximum.accumulate(np.flip(mpre)))
    x = np.linspace(0, 1, 101)
    return (getattr(np, "trapezoid", None) or np.trapz)(np.interp(x, mrec, mpre), x)


def ap_per_class(tp, conf, pred_cls, target_cls, eps=1e-16):
    """Returns dict(tp, fp, p, r, f1, ap (nc,10), classes, p_curve, r_curve, f1_curve)."""
    i = np.argsort(-conf)
    tp, conf, pred_cls = tp[i], conf[i], pred_cls[i]
    classes, nt = np.unique(target_cls, return_counts=True)
    nc = classes.shape[0]
    x = np.linspace(0, 1, 1000)
    ap, p_curve, r_curve = np.zeros((nc, tp.shape[1])), np.zeros((nc, 1000)), np.zeros((nc, 1000))
    for ci, c in enumerate(classes):
        sel = pred_cls == c
        n_l, n_p = nt[ci], sel.sum()
        if n_p == 0 or n_l == 0:
            continue
        fpc = (1 - tp[sel]).cumsum(0)
        tpc = tp[sel].cumsum(0)
        recall = tpc / (n_l + eps)
        r_curve[ci] = np.interp(-x, -conf[sel], recall[:, 0], left=0)
        precision = tpc / (tpc + fpc)
        p_curve[ci] = np.interp(-x, -conf[sel], precision[:, 0], left=1)
        for j in range(tp.shape[1]):
            ap[ci, j] = compute_ap(recall[:, j], precision[:, j])
    f1_curve = 2 * p_curve * r_curve / (p_curve + r_curve + eps)
    k = smooth(f1_curve.mean(0), 0.1).argmax()
    p, r, f1 = p_curve[:, k], r_curve[:, k], f1_curve[:, k]
    tpc_ = (r * nt).round()
    fp = (tpc_ / (p + eps) - tpc_).round()
    return dict(tp=tpc_, fp=fp, p=p, r=r, f1=f1, ap=ap, classes=classes.astype(int), p_curve=p_curve, r_curve=r_curve,
                f1_curve=f1_curve)


def mean_results(res):
    """(mp, mr, map50, map) of DetMetrics.mean_results; empty statistics give zeros."""
    if len(res["ap"]) == 0:
        return np.zeros(4)
    return np.array([res["p"].mean(), res["r"].mean(), res["ap"][:, 0].mean(), res["ap"].mean()])


def fitness(res):
    """Metric.fitness: weights [0, 0, 0.1, 0.9] over (mp, mr, map50, map)."""
    return float((mean_results(res) * np.array([0.0, 0.0, 0.1, 0.9])).sum())


def xywhn_to_xyxy(b, w, h):
    """ops.xywh2xyxy (utils/ops.py) followed by the * tensor(imgsz)[[1,0,1,0]] of _prepare_batch, fp32."""
    b = np.asarray(b, np.float32).reshape(-1, 4)
    half_w, half_h = b[:, 2] / np.float32(2), b[:, 3] / np.float32(2)
    out = np.stack([b[:, 0] - half_w, b[:, 1] - half_h, b[:, 0] + half_w, b[:, 1] + half_h], 1)
    return out * np.array([w, h, w, h], np.float32)


def scale_boxes(img1_shape, boxes, img0_shape, ratio_pad=None):
    """utils/ops.py scale_boxes (padding=True, xyxy) + clip_boxes, fp32; returns a new array."""
    if ratio_pad is None:
        gain = min(img1_shape[0] / img0_shape[0], img1_shape[1] / img0_shape[1])
        pad = (round((img1_shape[1] - img0_shape[1] * gain) / 2 - 0.1), round((img1_shape[0] - img0_shape[0] * gain) / 2 - 0.1))
    else:
        gain, pad = ratio_pad[0][0], ratio_pad[1]
    b = np.array(boxes, np.float32).reshape(-1, 4)
    b[:, [0, 2]] -= np.float32(pad[0])
    b[:, [1, 3]] -= np.float32(pad[1])
    b /= np.float32(gain)
    b[:, [0, 2]] = b[:, [0, 2]].clip(0, img0_shape[1])
    b[:, [1, 3]] = b[:, [1, 3]].clip(0, img0_shape[0])
    return b


def validate_batch(preds, batch, imgsz=(640, 640), geometry=None):
    """DetectionValidator.update_metrics for one batch (models/yolo/detect/val.py:117-161): preds = list of (n,6) xyxy/conf/cls
    arrays in network-input pixels; geometry = per-image (ori_shape, ratio_pad) (None: identity); returns the four stat lists."""
    stats = dict(tp=[], conf=[], pred_cls=[], target_cls=[])
    for si, pred in enumerate(preds):
        ori_shape, ratio_pad = geometry[si] if geometry is not None else (imgsz, ((1.0, 1.0), (0.0, 0.0)))
        idx = np.asarray(batch["batch_idx"]) == si
        tcls = np.asarray(batch["cls"]).reshape(-1)[idx]
        tbox = scale_boxes(imgsz, xywhn_to_xyxy(np.asarray(batch["bboxes"])[idx], imgsz[1], imgsz[0]), ori_shape, ratio_pad)
        pred = np.array(pred, np.float32).reshape(-1, 6)
        if len(pred) == 0:
            if len(tcls):
                stats["tp"].append(np.zeros((0, len(IOUV)), bool)); stats["conf"].append(np.zeros(0, np.float32))
                stats["pred_cls"].append(np.zeros(0, np.float32)); stats["target_cls"].append(tcls)
            continue
        pred[:, :4] = scale_boxes(imgsz, pred[:, :4], ori_shape, ratio_pad)
        tp = np.zeros((len(pred), len(IOUV)), bool)
        if len(tcls):
            tp = match_predictions(pred[:, 5], tcls, box_iou(tbox, pred[:, :4]))
        stats["tp"].append(tp); stats["conf"].append(pred[:, 4]); stats["pred_cls"].append(pred[:, 5]); stats["target_cls"].append(tcls)
    return stats
