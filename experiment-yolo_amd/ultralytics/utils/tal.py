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
    """reference utils/tal.py:13-88.  ``forward(pd_scores (B,A,nc) sigmoid, pd_bboxes (B,A,4) xyxy pixels, anc_points (A,2) pixels,
    gt_labels (B,n,1), gt_bboxes (B,n,4), mask_gt (B,n,1))`` -> (target_labels, target_bboxes, target_scores, fg_mask,
    target_gt_idx), computed by the assignment kernels of the loss (``dy_tal_assign``: tal_topk / scatter / resolve / scores).
    Only the configuration the reference itself builds (utils/loss.py:311: topk 10, alpha 0.5, beta 6.0) is on the HIP path.
    Anchors must be the level-major, row-major grid ``make_anchors`` produces (levels and strides are read back from anc_points)."""

    def __init__(self, topk=13, num_classes=80, alpha=1.0, beta=6.0, eps=1e-9):
        self.topk, self.num_classes, self.bg_idx, self.alpha, self.beta, self.eps = topk, num_classes, num_classes, alpha, beta, eps

    @staticmethod
    def _levels(anc_points):
        """(A,2) anchor centres in pixels -> [(H, W, stride)] (make_anchors: per level a row-major grid of (i + 0.5) * stride)."""
        pts = anc_points.detach().float().cpu()
        out, i, A = [], 0, pts.shape[0]
        while i < A:
            s = float(pts[i, 0]) * 2.0
            row = pts[i:, 1] == pts[i, 1]
            w = int(row.int().cumprod(0).sum())                       # anchors sharing the first row's y
            xs = pts[i:i + w, 0]
            if s <= 0 or w < 1 or not torch.allclose(xs, (torch.arange(w) + 0.5) * s):
                raise ValueError("anc_points is not a make_anchors grid (level-major, row-major, offset 0.5)")
            h = 0
            while i + h * w < A and float(pts[i + h * w, 0]) == float(pts[i, 0]) and abs(float(pts[i + h * w, 1]) - (h + 0.5) * s) < 1e-3 * s:
                h += 1
            out.append((h, w, s))
            i += h * w
        return out

    @torch.no_grad()
    def forward(self, pd_scores, pd_bboxes, anc_points, gt_labels, gt_bboxes, mask_gt):
        import ctypes as C
        from ..hip import check, lib
        if (self.topk, self.alpha, self.beta) != (10, 0.5, 6.0):
            raise NotImplementedError("the HIP assignment kernels implement the reference's own configuration (utils/loss.py:311): "
                                      f"topk=10, alpha=0.5, beta=6.0; got topk={self.topk}, alpha={self.alpha}, beta={self.beta}")
        dev = pd_scores.device
        if dev.type != "cuda":
            raise RuntimeError("TaskAlignedAssigner runs on the GPU only (no CPU fallback by design)")
        B, A, nc = pd_scores.shape
        n = gt_bboxes.shape[1]
        if n == 0:  # reference :58-67
            return (torch.full_like(pd_scores[..., 0], self.bg_idx), torch.zeros_like(pd_bboxes), torch.zeros_like(pd_scores),
                    torch.zeros_like(pd_scores[..., 0]), torch.zeros_like(pd_scores[..., 0]))
        levels = self._levels(anc_points)
        if sum(h * w for h, w, _ in levels) != A or len(levels) > 4:
            raise ValueError("anc_points does not describe the (<= 4) levels of pd_scores")
        ncp = (nc + 7) // 8 * 8
        scores, a0, st = [], 0, torch.empty(A, dtype=torch.float32)
        for h, w, s in levels:  # re-lay pd_scores per level as (B,H,W,ncp) -- plumbing for a caller holding reference-format tensors
            t = torch.zeros((B, h, w, ncp), dtype=torch.float32, device=dev)
            t[..., :nc] = pd_scores[:, a0:a0 + h * w].float().reshape(B, h, w, nc)
            scores.append(t)
            st[a0:a0 + h * w] = s
            a0 += h * w
        boxes_grid = (pd_bboxes.float() / st.to(dev)[None, :, None]).contiguous()
        labels = gt_labels.reshape(B, n).to(torch.int32).contiguous()
        gtb = gt_bboxes.float().reshape(B, n, 4).contiguous()
        mask = mask_gt.reshape(B, n).to(torch.int32).contiguous()
        L = lib()
        ws = torch.empty(L.dy_loss_workspace_bytes(B, A, n), dtype=torch.uint8, device=dev)
        asg = torch.empty((B, A), dtype=torch.int32, device=dev)
        ts = torch.empty((B, A), dtype=torch.float32, device=dev)
        nl = len(levels)
        ptrs = (C.c_void_p * 4)(*[t.data_ptr() for t in scores], *([None] * (4 - nl)))
        Hs, Ws = (C.c_int * 4)(*[l[0] for l in levels], *([0] * (4 - nl))), (C.c_int * 4)(*[l[1] for l in levels], *([0] * (4 - nl)))
        Ss = (C.c_float * 4)(*[l[2] for l in levels], *([0.0] * (4 - nl)))
        check(L.dy_tal_assign(ptrs, Hs, Ws, Ss, nl, B, nc, ncp, n, boxes_grid.data_ptr(), labels.data_ptr(), gtb.data_ptr(), mask.data_ptr(),
                              asg.data_ptr(), ts.data_ptr(), ws.data_ptr(), torch.cuda.current_stream(dev).cuda_stream), "dy_tal_assign")
        # the reference's return layout (get_targets :163-202): background anchors carry gt 0 of their image
        fg_mask = asg >= 0
        target_gt_idx = asg.clamp(min=0).long()
        flat = target_gt_idx + torch.arange(B, device=dev)[:, None] * n
        target_labels = gt_labels.long().flatten()[flat].clamp_(0)
        target_bboxes = gtb.view(-1, 4)[flat].to(gt_bboxes.dtype)
        target_scores = torch.zeros((B, A, self.num_classes), dtype=torch.int64, device=dev)
        target_scores.scatter_(2, target_labels.unsqueeze(-1), 1)
        target_scores = torch.where(fg_mask[:, :, None], target_scores, 0) * ts.unsqueeze(-1).to(pd_scores.dtype)
        return target_labels, target_bboxes, target_scores, fg_mask.bool(), target_gt_idx

    __call__ = forward
