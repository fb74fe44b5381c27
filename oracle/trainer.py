"""Oracle (test infrastructure): arithmetic of one reference training step on CPU.

Restates engine/trainer.py:784-815 (warm-up, forward, loss*world, backward), :949-957 (unscale, clip,
step, zero_grad, EMA), :1115-1180 (parameter groups, SGD-nesterov / AdamW) and
utils/torch_utils.py:431-458 (ModelEMA) over the functional oracle model.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field

import numpy as np
import torch

from . import nn as onn
from .graph import Graph, is_param
from .loss import LossState, detection_loss


def param_groups(names):
    """engine/trainer.py:1146-1154 -> (bias, decayed weights, norm weights); .dfl is frozen (:670)."""
    g_bias, g_w, g_bn = [], [], []
    for n in names:
        if not is_param(n):
            continue
        if "bias" in n:
            g_bias.append(n)
        elif n.endswith(".bn.weight") or n.endswith(".conv.1.weight"):
            g_bn.append(n)
        else:
            g_w.append(n)
    return g_bias, g_w, g_bn


@dataclass
class Hyp:
    """The cfg/default.yaml keys the step reads."""
    lr0: float = 0.01
    lrf: float = 0.01
    momentum: float = 0.937
    weight_decay: float = 0.0005
    warmup_epochs: float = 3.0
    warmup_momentum: float = 0.8
    warmup_bias_lr: float = 0.1
    nbs: int = 64
    box: float = 7.5
    cls: float = 0.5
    dfl: float = 1.5
    epochs: int = 100
    optimizer: str = "SGD"
    cos_lr: bool = False


@dataclass
class TrainState:
    g: Graph
    sd: dict
    hyp: Hyp
    batch_size: int
    nb: int  # batches per epoch
    world_size: int = 1
    loss_state: LossState = field(default_factory=LossState)
    bufs: dict = field(default_factory=dict)  # momentum buffers / adam moments
    ema: dict | None = None
    ema_updates: int = 0
    ni: int = 0
    last_opt_step: int = -1
    accum: dict = field(default_factory=dict)
    adam_t: int = 0

    def __post_init__(self):
        self.ema = {k: v.clone() for k, v in self.sd.items()}
        self.groups = param_groups(self.sd.keys())
        self.accumulate = max(round(self.hyp.nbs / self.batch_size), 1)
        self.wd = self.hyp.weight_decay * self.batch_size * self.accumulate / self.hyp.nbs
        self.nw = max(round(self.hyp.warmup_epochs * self.nb), 100) if self.hyp.warmup_epochs > 0 else -1
        self.group_lr = [self.hyp.lr0] * 3
        self.mom = self.hyp.momentum

    def lf(self, epoch):
        h = self.hyp
        if h.cos_lr:
            return max((1 - math.cos(epoch * math.pi / h.epochs)) / 2, 0) * (h.lrf - 1) + 1
        return max(1 - epoch / h.epochs, 0) * (1.0 - h.lrf) + h.lrf

    def warmup(self, epoch):
        """engine/trainer.py:784-793."""
        ni, nw, h = self.ni, self.nw, self.hyp
        if ni <= nw:
            xi = [0, nw]
            self.accumulate = max(1, int(np.interp(ni, xi, [1, h.nbs / self.batch_size]).round()))
            for j in range(3):
                self.group_lr[j] = float(np.interp(ni, xi, [h.warmup_bias_lr if j == 0 else 0.0, h.lr0 * self.lf(epoch)]))
            self.mom = float(np.interp(ni, xi, [h.warmup_momentum, h.momentum]))


def train_step(ts: TrainState, batch, epoch=0):
    """One iteration of the hot loop; returns (loss, loss_items, grad_norm or None)."""
    ts.warmup(epoch)
    names = [n for grp in ts.groups for n in grp if ".dfl." not in n]  # .dfl frozen: grad stays None
    leaves = {n: ts.sd[n].detach().requires_grad_(True) for n in names}
    sd = dict(ts.sd)
    sd.update(leaves)
    feats = onn.forward(ts.g, sd, batch["img"], training=True)
    h = ts.hyp
    loss, items = detection_loss(feats, batch, ts.g.strides, ts.g.nc, (h.box, h.cls, h.dfl), ts.loss_state)
    loss = loss * ts.world_size if ts.world_size > 1 else loss
    grads = torch.autograd.grad(loss, [leaves[n] for n in names], allow_unused=True)
    for n, gr in zip(names, grads):
        if gr is not None:
            ts.accum[n] = ts.accum.get(n, 0) + gr
    gnorm = None
    if ts.ni - ts.last_opt_step >= ts.accumulate:
        gnorm = optimizer_step(ts)
        ts.last_opt_step = ts.ni
    ts.ni += 1
    return float(loss.detach()), items, gnorm


@torch.no_grad()
def optimizer_step(ts: TrainState):
    """engine/trainer.py:949-957 with torch.optim.SGD(nesterov=True) / AdamW semantics."""
    grads = ts.accum
    total = torch.sqrt(sum((g.float() ** 2).sum() for g in grads.values())) if grads else torch.tensor(0.0)
    coef = torch.clamp(10.0 / (total + 1e-6), max=1.0)  # clip_grad_norm_(max_norm=10)
    adam = ts.hyp.optimizer in ("Adam", "AdamW")
    if adam:
        ts.adam_t += 1
    for gi, (grp, wd) in enumerate(zip(ts.groups, (0.0, ts.wd, 0.0))):
        lr = ts.group_lr[gi]
        for n in grp:
            if n not in grads:
                continue
            p, g = ts.sd[n], grads[n] * coef
            if adam:  # betas=(momentum, 0.999), eps 1e-8, decoupled decay for AdamW
                m, v = ts.bufs.setdefault(n, (torch.zeros_like(p), torch.zeros_like(p)))
                if ts.hyp.optimizer == "AdamW":
                    p.mul_(1 - lr * wd)
                elif wd:
                    g = g + wd * p
                b1 = ts.hyp.momentum  # Adam groups carry no "momentum" key, so warm-up never touches beta1 (:792)
                m.mul_(b1).add_(g, alpha=1 - b1)
                v.mul_(0.999).addcmul_(g, g, value=0.001)
                bc1, bc2 = 1 - b1 ** ts.adam_t, 1 - 0.999 ** ts.adam_t
                p.addcdiv_(m, (v.sqrt() / math.sqrt(bc2)).add_(1e-8), value=-lr / bc1)
            else:
                if wd:
                    g = g + wd * p
                if n not in ts.bufs:
                    ts.bufs[n] = g.clone()
                else:
                    ts.bufs[n].mul_(ts.mom).add_(g)
                p.sub_(lr * (g + ts.mom * ts.bufs[n]))
    ts.accum = {}
    # ModelEMA.update, utils/torch_utils.py:447-458
    ts.ema_updates += 1
    d = 0.9999 * (1 - math.exp(-ts.ema_updates / 2000))
    for k, v in ts.ema.items():
        if v.dtype.is_floating_point:
            v.mul_(d).add_((1 - d) * ts.sd[k])
    return float(total)
