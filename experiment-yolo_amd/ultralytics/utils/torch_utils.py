"""Device / seeding / EMA helpers (drop-in for the pieces of reference utils/torch_utils.py used by the entry scripts)."""
from __future__ import annotations

import math
import os
import random
from copy import deepcopy

import numpy as np
import torch


def select_device(device="", batch=0, newline=False, verbose=True):
    """'0' / 'cuda:0' / 0 -> torch.device; CPU requests raise because the hot path is GPU-only."""
    if isinstance(device, torch.device):
        return device
    d = "" if device is None else str(device).lower().replace("cuda:", "").strip()
    if d in ("cpu", "mps"):
        raise ValueError("the MI355X DEAL-YOLO path runs on GPUs only (device='cpu' is served by the reference implementation)")
    ids = [int(x) for x in d.split(",") if x != ""] or [0]
    if len(ids) > 1 and "LOCAL_RANK" not in os.environ:
        raise ValueError(f"device='{device}' lists {len(ids)} GPUs but this process is not a rank of a distributed launch: "
                         "YOLO.train(device='0,1,...') re-launches itself under torch.distributed.run; other entry points take one GPU")
    if not torch.cuda.is_available():
        raise ValueError(f"Invalid device '{device}' requested: no GPU visible")
    if "LOCAL_RANK" in os.environ:  # one rank per GPU under torch.distributed.run: the launcher narrowed the visible devices
        return torch.device("cuda", 0 if os.environ.get("DY_REHEARSE_ON_ONE_GPU") == "1" else int(os.environ["LOCAL_RANK"]))
    return torch.device("cuda", ids[0])


def init_seeds(seed=0, deterministic=False):
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)


def de_parallel(model):
    return model.module if hasattr(model, "module") else model


class ModelEMA:
    """Exponential moving average of parameters and float buffers (reference utils/torch_utils.py:431-464).  The averages
    live in the StepPlan's flat device buffers and are updated inside the optimizer kernel; ``.ema`` materialises a model
    carrying them (for validation / checkpoints)."""

    def __init__(self, plan, decay=0.9999, tau=2000, updates=0):
        self.plan, self.decay0, self.tau = plan, decay, tau
        self.enabled = True

    @property
    def updates(self):
        return self.plan.ema_updates

    def decay(self, x):
        return self.decay0 * (1 - math.exp(-x / self.tau))

    def state_dict(self):
        """EMA values under the model's state_dict keys (parameters and float buffers; integer buffers are copied)."""
        plan, rt = self.plan, self.plan.rt
        out = {}
        params = dict(plan.model.named_parameters())
        fb_off, o = {}, 0
        for mname, mod in plan.model.named_modules():
            for bname, b in mod.named_buffers(recurse=False):
                if b is not None and b.dtype.is_floating_point:
                    fb_off[f"{mname}.{bname}" if mname else bname] = (o, b.numel())
                    o += (b.numel() + 7) // 8 * 8
        for k, v in plan.model.state_dict().items():
            if k in params:
                off = rt.param_off[k]
                out[k] = plan.ema[off:off + v.numel()].view(v.shape).clone()
            elif k in fb_off:
                off, n = fb_off[k]
                out[k] = plan.ema_b[off:off + n].view(v.shape).clone()
            else:
                out[k] = v.clone()
        return out

    @property
    def ema(self):
        """A fresh eval-mode model carrying the averaged weights."""
        from ..nn.tasks import DetectionModel
        m = DetectionModel(deepcopy(self.plan.model.yaml), verbose=False)
        m.load_state_dict({k: v.cpu() for k, v in self.state_dict().items()}, strict=True)
        return m.eval()
