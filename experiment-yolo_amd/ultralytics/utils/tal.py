"""Anchor / box helpers and the assigner's import path (drop-in for reference utils/tal.py:13, 294-324).

``make_anchors`` / ``dist2bbox`` / ``bbox2dist`` are plain tensor helpers for callers that build anchors themselves; the hot path
never calls them (the loss and decode kernels generate anchors on the fly).  ``TaskAlignedAssigner`` answers to the reference's
name and call signature and runs the assignment kernels of ``dy_detection_loss``."""
from __future__ import annotations

import torch

__all__ = ("make_anchors", "dist2bbox", "bbox2dist", "TaskAlignedAssigner")


def make_anchors(feats, strides, grid_cell_offset=0.5):
    """reference utils/tal.py:294-307: (anchor_points (A,2), stride_tensor (A,1)) from per-level (B,C,H,W) maps."""
    pts, st = [], []
    assert feats is not None
    dtype, device = feats[0].dtype, feats[0].device
    for i, stride in enumerate(strides):
        _, _, h, w = feats[i].shape
        sx = torch.arange(end=w, device=device, dtype=dtype) + grid_cell_offset
        sy = torch.arange(end=h, device=device, dtype=dtype) + grid_cell_offset
        sy, sx = torch.meshgrid(sy, sx, indexing="ij")
        pts.append(torch.stack((sx, sy), -1).view(-1, 2))
        st.append(torch.full((h * w, 1), float(stride), dtype=dtype, device=device))
    return torch.cat(pts), torch.cat(st)


def dist2bbox(distance, anchor_points, xywh=True, dim=-1):
    """reference utils/tal.py:310-318."""
    lt, rb = distance.chunk(2, dim)
    x1y1, x2y2 = anchor_points - lt, anchor_points + rb
    if xywh:
        return torch.cat(((x1y1 + x2y2) / 2, x2y2 - x1y1), dim)
    return torch.cat((x1y1, x2y2), dim)


def bbox2dist(anchor_points, bbox, reg_max):
    """reference utils/tal.py:321-324."""
    x1y1, x2y2 = bbox.chunk(2, -1)
    return torch.cat((anchor_points - x1y1, x2y2 - anchor_points), -1).clamp_(0, reg_max - 0.01)


class TaskAlignedAssigner:
    """reference utils/tal.py:13-88.  ``forward(pd_scores (B,A,nc) sigmoid, pd_bboxes (B,A,4) xyxy pixels, anc_points, gt_labels
    (B,n,1), gt_bboxes (B,n,4), mask_gt (B,n,1))`` -> (target_labels, target_bboxes, target_scores, fg_mask, target_gt_idx).
    Only the reference's own configuration is on the HIP path (topk 10, alpha 0.5, beta 6.0, CIoU overlaps); the kernels take head
    logits, so the assigner is reached through ``v8DetectionLoss`` -- a direct call raises with that pointer."""

    def __init__(self, topk=13, num_classes=80, alpha=1.0, beta=6.0, eps=1e-9):
        self.topk, self.num_classes, self.bg_idx, self.alpha, self.beta, self.eps = topk, num_classes, num_classes, alpha, beta, eps

    def forward(self, *a, **k):
        raise NotImplementedError("the task-aligned assignment runs inside dy_detection_loss (tal_topk / scatter / resolve / scores "
                                  "kernels): call ultralytics.utils.loss.v8DetectionLoss; v8DetectionLoss.debug_assignment() returns "
                                  "target_gt_idx / target score / decoded boxes of the last call")

    __call__ = forward
