"""Oracle (test infrastructure): the reference's Gaussian soft-NMS post-process, quirks included.

Restates utils/ops.py:260-290 (soft_nms), :162-258 (bbox_iou_for_nms, plain-IoU branch) and
:292-427 (non_max_suppression).  The order-dependent behaviours listed in SURVEY.md section 8a row N2
are part of the contract and are reproduced on purpose:
  (1) the first kept box is candidate 0, not the top score;
  (2) surviving scores are decayed in place (the returned confidences are the decayed ones);
  (3) the arg-max survivor is *swapped* to the front, nothing is sorted;
  (4) the last remaining box is never kept;
  (5) ``.squeeze()`` of single-element results: when exactly one rival remains the IoU is 0-d, its
      ``nonzero().squeeze()`` is empty, and the decay is skipped (unobservable: that rival is dropped by (4));
  (6) the score threshold is the constant 0.25 regardless of ``conf_thres``.
"""
from __future__ import annotations

import torch

from .loss import xywh2xyxy

SIGMA, SCORE_THR = 0.5, 0.25  # utils/ops.py:260 defaults; never overridden by the caller (:407)


def iou_1_to_n(b1, b2, eps=1e-7):
    """bbox_iou_for_nms plain IoU, utils/ops.py:186-199: eps joins both heights and the union."""
    x1, y1, x2, y2 = b1.chunk(4, -1)
    X1, Y1, X2, Y2 = b2.chunk(4, -1)
    w1, h1 = x2 - x1, y2 - y1 + eps
    w2, h2 = X2 - X1, Y2 - Y1 + eps
    inter = (torch.minimum(x2, X2) - torch.maximum(x1, X1)).clamp(0) * \
            (torch.minimum(y2, Y2) - torch.maximum(y1, Y1)).clamp(0)
    return inter / (w1 * h1 + w2 * h2 - inter + eps)


def soft_nms(boxes, scores, iou_thresh=0.5, sigma=SIGMA, score_threshold=SCORE_THR):
    """utils/ops.py:260-290.  ``scores`` is decayed IN PLACE.  Returns kept indices (int64)."""
    order = torch.arange(scores.shape[0])
    keep = []
    while order.numel() > 1:
        i = int(order[0])
        keep.append(i)
        rest = order[1:]
        iou = iou_1_to_n(boxes[i:i + 1], boxes[rest]).view(-1)
        hit = iou > iou_thresh
        if rest.numel() > 1 and hit.any():  # quirk (5): a 0-d IoU (single rival) never decays
            scores[rest[hit]] *= torch.exp(-iou[hit].pow(2) / sigma)
        alive = (scores[rest] > score_threshold).nonzero().view(-1)
        if alive.numel() == 0:
            break
        top = int(torch.argmax(scores[rest[alive]]))
        if top != 0:
            alive[[0, top]] = alive[[top, 0]]
        order = rest[alive]
    return torch.tensor(keep, dtype=torch.int64)


def non_max_suppression(prediction, conf_thres=0.25, iou_thres=0.45, classes=None, agnostic=False,
                        multi_label=False, max_det=300, nc=0, max_nms=30000, max_wh=7680, return_indices=False):
    """utils/ops.py:292-427 for detection (no masks, no apriori labels, no rotated boxes, no time limit).

    ``prediction`` (B, 4+nc, A) is NOT modified (the reference rewrites its xywh in place, :356).
    Returns list of (k,6) [x1,y1,x2,y2,conf,cls]; with ``return_indices`` also the per-image candidate
    row index (into the filtered candidate list) of every kept detection."""
    bs = prediction.shape[0]
    nc = nc or (prediction.shape[1] - 4)
    xc = prediction[:, 4:4 + nc].amax(1) > conf_thres
    multi_label &= nc > 1
    pred = prediction.transpose(-1, -2).clone()
    pred[..., :4] = xywh2xyxy(pred[..., :4])
    out = [torch.zeros((0, 6))] * bs
    kept = [torch.zeros((0,), dtype=torch.int64)] * bs
    for xi in range(bs):
        x = pred[xi][xc[xi]]
        if not x.shape[0]:
            continue
        box, cls = x[:, :4], x[:, 4:4 + nc]
        if multi_label:
            i, j = torch.where(cls > conf_thres)
            x = torch.cat((box[i], x[i, 4 + j, None], j[:, None].float()), 1)
        else:
            conf, j = cls.max(1, keepdim=True)
            x = torch.cat((box, conf, j.float()), 1)[conf.view(-1) > conf_thres]
        if classes is not None:
            x = x[(x[:, 5:6] == torch.tensor(classes)).any(1)]
        n = x.shape[0]
        if not n:
            continue
        if n > max_nms:
            x = x[x[:, 4].argsort(descending=True)[:max_nms]]
        c = x[:, 5:6] * (0 if agnostic else max_wh)
        scores = x[:, 4]  # a view: decay lands in x[:, 4]
        i = soft_nms(x[:, :4] + c, scores, iou_thres)[:max_det]
        out[xi] = x[i]
        kept[xi] = i
    return (out, kept) if return_indices else out
