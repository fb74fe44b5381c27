"""Oracle (test infrastructure): detection loss -- TAL assigner + CIoU | WIoU-v3 + NWD + DFL + BCE.

Restates utils/loss.py:187-250, :294-457, utils/tal.py:13-324 and utils/metrics.py:75-126, :540-645 as
stateless functions (the WIoU running mean is passed in and returned).
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import torch
import torch.nn.functional as F

from .graph import REG_MAX
from .nn import make_anchors

TOPK, ALPHA, BETA, TAL_EPS = 10, 0.5, 6.0, 1e-9  # utils/loss.py:316
WIOU_MOMENTUM, WIOU_ALPHA, WIOU_DELTA = 1e-2, 1.7, 2.7  # utils/metrics.py:573-575
NWD_CONSTANT = 12.8  # utils/metrics.py:540


def xywh2xyxy(x):
    """utils/ops.py:527-546."""
    xy, wh = x[..., :2], x[..., 2:4] / 2
    return torch.cat((xy - wh, xy + wh), -1)


def ciou(b1, b2, eps=1e-7):
    """bbox_iou(xywh=False, CIoU=True), utils/metrics.py:75-126.  Shapes (...,4) -> (...,1)."""
    x1, y1, x2, y2 = b1.chunk(4, -1)
    X1, Y1, X2, Y2 = b2.chunk(4, -1)
    w1, h1 = x2 - x1, y2 - y1 + eps
    w2, h2 = X2 - X1, Y2 - Y1 + eps
    inter = (torch.minimum(x2, X2) - torch.maximum(x1, X1)).clamp(0) * \
            (torch.minimum(y2, Y2) - torch.maximum(y1, Y1)).clamp(0)
    union = w1 * h1 + w2 * h2 - inter + eps
    iou = inter / union
    cw = torch.maximum(x2, X2) - torch.minimum(x1, X1)
    ch = torch.maximum(y2, Y2) - torch.minimum(y1, Y1)
    c2 = cw ** 2 + ch ** 2 + eps
    rho2 = ((X1 + X2 - x1 - x2) ** 2 + (Y1 + Y2 - y1 - y2) ** 2) / 4
    v = (4 / math.pi ** 2) * (torch.atan(w2 / h2) - torch.atan(w1 / h1)).pow(2)
    with torch.no_grad():
        alpha = v / (v - iou + (1 + eps))
    return iou - (rho2 / c2 + v * alpha)


def nwd(pred, target, eps=1e-7, constant=NWD_CONSTANT):
    """wasserstein_loss, utils/metrics.py:540-565 -> (n,1)."""
    x1, y1, x2, y2 = pred.chunk(4, -1)
    X1, Y1, X2, Y2 = target.chunk(4, -1)
    w1, h1 = x2 - x1, y2 - y1 + eps
    w2, h2 = X2 - X1, Y2 - Y1 + eps
    cd = ((x1 + w1 / 2) - (X1 + w2 / 2)) ** 2 + ((y1 + h1 / 2) - (Y1 + h2 / 2)) ** 2 + eps
    wh = ((w1 - w2) ** 2 + (h1 - h2) ** 2) / 4
    return torch.exp(-torch.sqrt(cd + wh) / constant)


def wiou_v3(pred, target, iou_mean, update=True):
    """WiseIouLoss(ltype='WIoU', monotonous=False).forward, utils/metrics.py:591-645.

    Returns (loss (n,), new_iou_mean).  ``iou_mean`` is the running mean of L_IoU = 1 - IoU (init 1.0)."""
    p_xy, p_wh = (pred[..., :2] + pred[..., 2:4]) / 2, pred[..., 2:4] - pred[..., :2]
    t_xy, t_wh = (target[..., :2] + target[..., 2:4]) / 2, target[..., 2:4] - target[..., :2]
    mn, mx = torch.minimum(pred, target), torch.maximum(pred, target)
    wh_inter = torch.relu(mn[..., 2:4] - mx[..., :2])
    s_inter = wh_inter.prod(-1)
    s_union = p_wh.prod(-1) + t_wh.prod(-1) - s_inter
    wh_box = mx[..., 2:4] - mn[..., :2]
    l2_box = wh_box.square().sum(-1)
    l2_center = (p_xy - t_xy).square().sum(-1)
    l_iou = 1 - s_inter / s_union
    if update:  # ``self.training`` is always True on the criterion (WIoU state)
        iou_mean = iou_mean * (1 - WIOU_MOMENTUM) + WIOU_MOMENTUM * l_iou.detach().mean()
    loss = torch.exp(l2_center / l2_box.detach()) * l_iou
    beta = l_iou.detach() / iou_mean
    loss = loss * (beta / (WIOU_DELTA * torch.pow(WIOU_ALPHA, beta - WIOU_DELTA)))
    return loss, iou_mean


# ----------------------------------------------------------------------------- task-aligned assigner
@dataclass
class Assignment:
    target_labels: torch.Tensor
    target_bboxes: torch.Tensor
    target_scores: torch.Tensor
    fg_mask: torch.Tensor
    target_gt_idx: torch.Tensor


@torch.no_grad()
def tal_assign(pd_scores, pd_bboxes, anc, gt_labels, gt_bboxes, mask_gt, nc, topk=TOPK):
    """TaskAlignedAssigner.forward, utils/tal.py:39-88 (+ helpers :90-258)."""
    bs, na = pd_scores.shape[:2]
    nm = gt_bboxes.shape[1]
    if nm == 0:  # :59-67
        z = torch.zeros_like(pd_scores[..., 0])
        return Assignment(torch.full_like(z, nc), torch.zeros_like(pd_bboxes), torch.zeros_like(pd_scores), z.bool(), z)
    # select_candidates_in_gts :213-229
    lt, rb = gt_bboxes.view(-1, 1, 4).chunk(2, 2)
    deltas = torch.cat((anc[None] - lt, rb - anc[None]), 2).view(bs, nm, na, -1)
    mask_in = deltas.amin(3).gt(TAL_EPS).to(gt_bboxes.dtype)
    # get_box_metrics :102-121
    m = (mask_in * mask_gt).bool()
    overlaps = torch.zeros(bs, nm, na, dtype=pd_bboxes.dtype)
    scores = torch.zeros(bs, nm, na, dtype=pd_scores.dtype)
    bi = torch.arange(bs).view(-1, 1).expand(-1, nm)
    scores[m] = pd_scores[bi, :, gt_labels.squeeze(-1).long()][m]
    pb = pd_bboxes.unsqueeze(1).expand(-1, nm, -1, -1)[m]
    gb = gt_bboxes.unsqueeze(2).expand(-1, -1, na, -1)[m]
    overlaps[m] = ciou(gb, pb).squeeze(-1).clamp_(0)  # :123-125 (CIoU, clamped)
    align = scores.pow(ALPHA) * overlaps.pow(BETA)
    # select_topk_candidates :127-161
    _, idx = torch.topk(align, topk, dim=-1, largest=True)
    idx.masked_fill_(~mask_gt.expand(-1, -1, topk).bool(), 0)
    count = torch.zeros(align.shape, dtype=torch.int8)
    ones = torch.ones_like(idx[:, :, :1], dtype=torch.int8)
    for k in range(topk):
        count.scatter_add_(-1, idx[:, :, k:k + 1], ones)
    count.masked_fill_(count > 1, 0)
    mask_pos = count.to(align.dtype) * mask_in * mask_gt  # :98
    # select_highest_overlaps :232-258
    fg = mask_pos.sum(-2)
    if fg.max() > 1:
        multi = (fg.unsqueeze(1) > 1).expand(-1, nm, -1)
        best = overlaps.argmax(1)
        is_max = torch.zeros_like(mask_pos)
        is_max.scatter_(1, best.unsqueeze(1), 1)
        mask_pos = torch.where(multi, is_max, mask_pos).float()
        fg = mask_pos.sum(-2)
    tgi = mask_pos.argmax(-2)
    # get_targets :163-210
    flat = tgi + torch.arange(bs)[:, None] * nm
    labels = gt_labels.long().flatten()[flat].clamp_(0)
    tboxes = gt_bboxes.view(-1, 4)[flat]
    tscores = torch.zeros(bs, na, nc, dtype=torch.int64)
    tscores.scatter_(2, labels.unsqueeze(-1), 1)
    tscores = torch.where(fg[:, :, None].repeat(1, 1, nc) > 0, tscores, 0)
    # normalise :80-86
    align = align * mask_pos
    pos_align = align.amax(-1, keepdim=True)
    pos_ov = (overlaps * mask_pos).amax(-1, keepdim=True)
    norm = (align * pos_ov / (pos_align + TAL_EPS)).amax(-2).unsqueeze(-1)
    return Assignment(labels, tboxes, tscores * norm, fg.bool(), tgi)


def pack_targets(batch, bs, imgsz_wh):
    """v8DetectionLoss.preprocess, utils/loss.py:330-345: (n,6) rows -> (B, n_max, 5) [cls, xyxy px]."""
    t = torch.cat((batch["batch_idx"].view(-1, 1), batch["cls"].view(-1, 1), batch["bboxes"]), 1).float()
    if t.shape[0] == 0:
        return torch.zeros(bs, 0, 5)
    i = t[:, 0]
    counts = torch.stack([(i == j).sum() for j in range(bs)])
    out = torch.zeros(bs, int(counts.max()), 5)
    for j in range(bs):
        sel = i == j
        n = int(sel.sum())
        if n:
            out[j, :n] = t[sel, 1:]
    out[..., 1:5] = xywh2xyxy(out[..., 1:5] * imgsz_wh)
    return out


def df_loss(pred_dist, target):
    """BboxLoss._df_loss, utils/loss.py:236-250."""
    tl = target.long()
    tr = tl + 1
    wl = tr - target
    wr = 1 - wl
    return (F.cross_entropy(pred_dist, tl.view(-1), reduction="none").view(tl.shape) * wl
            + F.cross_entropy(pred_dist, tr.view(-1), reduction="none").view(tl.shape) * wr).mean(-1, keepdim=True)


@dataclass
class LossState:
    """Mutable criterion state: the toggles (utils/loss.py:194,197) and the WIoU running mean."""
    use_wiseiou: bool = False
    nwd_loss: bool = False
    iou_ratio: float = 0.5
    iou_mean: float = 1.0


def detection_loss(feats, batch, strides, nc, hyp=(7.5, 0.5, 1.5), state: LossState | None = None, detail=False):
    """v8DetectionLoss.compute_loss/__call__, utils/loss.py:356-457 + BboxLoss.forward :202-233.

    Returns (loss.sum()*B, loss_items[box, cls, dfl]) and, with ``detail``, the Assignment and
    target_scores_sum as well.  ``state.iou_mean`` is updated in place in WIoU mode."""
    state = state or LossState()
    no = nc + 4 * REG_MAX
    bs = feats[0].shape[0]
    cat = torch.cat([f.reshape(bs, no, -1) for f in feats], 2)
    pred_distri, pred_scores = cat.split((4 * REG_MAX, nc), 1)
    pred_scores = pred_scores.permute(0, 2, 1).contiguous()
    pred_distri = pred_distri.permute(0, 2, 1).contiguous()
    h0, w0 = feats[0].shape[2:]
    imgsz = torch.tensor([h0, w0], dtype=torch.float32) * strides[0]
    anchors, st = make_anchors([f.shape[2:] for f in feats], strides)
    targets = pack_targets(batch, bs, imgsz[[1, 0, 1, 0]])
    gt_labels, gt_bboxes = targets.split((1, 4), 2)
    mask_gt = gt_bboxes.sum(2, keepdim=True).gt(0).float()
    # bbox_decode :347-354
    b, a, c = pred_distri.shape
    proj = torch.arange(REG_MAX, dtype=torch.float32)
    d = pred_distri.view(b, a, 4, c // 4).softmax(3).matmul(proj)
    lt, rb = d.chunk(2, -1)
    pred_bboxes = torch.cat((anchors - lt, anchors + rb), -1)
    asg = tal_assign(pred_scores.detach().sigmoid(), pred_bboxes.detach() * st, anchors * st,
                     gt_labels, gt_bboxes, mask_gt, nc)
    tss = max(asg.target_scores.sum(), 1)
    loss = torch.zeros(3)
    loss[1] = F.binary_cross_entropy_with_logits(pred_scores, asg.target_scores, reduction="none").sum() / tss
    fg = asg.fg_mask
    if fg.sum():
        tb = asg.target_bboxes / st
        weight = asg.target_scores.sum(-1)[fg].unsqueeze(-1)
        if state.use_wiseiou:
            wl, new_mean = wiou_v3(pred_bboxes[fg], tb[fg], torch.as_tensor(state.iou_mean, dtype=torch.float32))
            state.iou_mean = float(new_mean)
            l_iou = (wl.unsqueeze(-1) * weight).sum() / tss
        else:
            l_iou = ((1.0 - ciou(pred_bboxes[fg], tb[fg])) * weight).sum() / tss
        if state.nwd_loss:
            l_nwd = ((1.0 - nwd(pred_bboxes[fg], tb[fg])) * weight).sum() / tss
            l_iou = state.iou_ratio * l_iou + (1 - state.iou_ratio) * l_nwd
        x1y1, x2y2 = tb.chunk(2, -1)  # bbox2dist, utils/tal.py:321-324
        ltrb = torch.cat((anchors - x1y1, x2y2 - anchors), -1).clamp_(0, REG_MAX - 1 - 0.01)
        l_dfl = (df_loss(pred_distri[fg].view(-1, REG_MAX), ltrb[fg]) * weight).sum() / tss
        loss[0], loss[2] = l_iou, l_dfl
    loss = loss * torch.tensor(hyp)
    out = (loss.sum() * bs, loss.detach())
    if detail:
        return out + (asg, float(tss))
    return out
