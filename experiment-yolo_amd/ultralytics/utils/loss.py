"""Detection criterion (drop-in for reference utils/loss.py:187-250, 294-457) issued as HIP kernels.

``v8DetectionLoss(model)(preds, batch)`` keeps the reference protocol and attribute names -- ``crit.bbox_loss.use_wiseiou``,
``crit.bbox_loss.nwd_loss``, ``crit.bbox_loss.iou_ratio`` toggle WIoU-v3 / NWD exactly as editing the two literals at
loss.py:194,197 does in the reference, and ``crit.bbox_loss.wiou_loss.iou_mean`` is the WIoU running mean -- but the
whole computation (target packing, DFL decode, task-aligned assignment, BCE/CIoU|WIoU/NWD/DFL and the gradients w.r.t.
the head logits) is one C-ABI call, ``dy_detection_loss``.
"""
from __future__ import annotations

import ctypes as C

import torch

from ..hip import DyLossArgs, check


class _WiouState:
    """Stands in for ``WiseIouLoss``: only its running mean is state (reference utils/metrics.py:573-589)."""
    momentum, alpha, delta = 1e-2, 1.7, 2.7

    def __init__(self, scalars):
        self._s = scalars

    @property
    def iou_mean(self):
        return self._s[4]

    @iou_mean.setter
    def iou_mean(self, v):
        self._s[4] = float(v)


class BboxLoss:
    """Toggle holder mirroring reference ``BboxLoss.__init__`` (utils/loss.py:189-200)."""

    def __init__(self, reg_max, use_dfl, scalars):
        self.reg_max, self.use_dfl = reg_max, use_dfl
        self.nwd_loss = False
        self.iou_ratio = 0.5
        self.use_wiseiou = False
        self.wiou_loss = _WiouState(scalars)


class v8DetectionLoss:
    def __init__(self, model):
        m = model.model[-1]
        h = getattr(model, "args", None)
        self.hyp = h
        self.box_gain = float(getattr(h, "box", 7.5)) if h is not None else 7.5
        self.cls_gain = float(getattr(h, "cls", 0.5)) if h is not None else 0.5
        self.dfl_gain = float(getattr(h, "dfl", 1.5)) if h is not None else 1.5
        self.stride, self.nc, self.no, self.reg_max = m.stride, m.nc, m.no, m.reg_max
        self.device = next(model.parameters()).device
        if self.device.type != "cuda":
            raise RuntimeError("v8DetectionLoss: the HIP loss kernels need the model on a GPU (no CPU fallback)")
        self.use_dfl = m.reg_max > 1
        self.scalars = torch.zeros(16, dtype=torch.float32, device=self.device)
        self.scalars[4] = 1.0  # WIoU iou_mean buffer initial value
        self.bbox_loss = BboxLoss(m.reg_max - 1, self.use_dfl, self.scalars)
        self.nmax = 0
        self._ws = None
        self._args = DyLossArgs()
        self._one = torch.ones(1, dtype=torch.float32, device=self.device)
        self._tgt = None
        self._ncount = None

    def clone_for_plan(self):
        """A criterion for a second recorded StepPlan on the same model (the ragged last batch of an epoch has its own launch
        list): own argument block, workspace and target buffers; the result scalars, the running WIoU mean and the mode toggles
        (``bbox_loss``) are THE SAME objects, as there is one criterion in the reference."""
        c = object.__new__(type(self))
        c.__dict__.update({k: v for k, v in self.__dict__.items() if k not in ("_pub", "_last")})
        c._args, c._ws, c._tgt, c._ncount = DyLossArgs(), None, None, None
        return c

    # ---- argument block ----------------------------------------------------------------------------------------
    def bind(self, ho, nmax, gscale=None):
        """Fill the DyLossArgs block for one HeadOut geometry; returns it (kept alive by self)."""
        from ..hip import lib
        L = lib()
        a = self._args
        a.nl, a.B, a.nc, a.ncp = len(ho.box), ho.box[0].shape[0], self.nc, ho.cls[0].shape[-1]
        a.nmax = nmax
        a.dbox_rows_only = a.box_from_input = 0  # a StepPlan trace switches them on AFTER binding (hip/train.py); every other caller is dense
        A = 0
        for l in range(a.nl):
            a.box[l], a.cls[l] = ho.box[l].data_ptr(), ho.cls[l].data_ptr()
            a.dbox[l] = ho.dbox[l].data_ptr() if ho.dbox is not None else 0
            a.dcls[l] = ho.dcls[l].data_ptr() if ho.dcls is not None else 0
            a.H[l], a.W[l], a.stride[l] = ho.box[l].shape[1], ho.box[l].shape[2], float(ho.strides[l])
            A += a.H[l] * a.W[l]
        a.img_w, a.img_h = a.W[0] * a.stride[0], a.H[0] * a.stride[0]
        a.gscale = (gscale if gscale is not None else self._one).data_ptr()
        a.scalars = self.scalars.data_ptr()
        need = L.dy_loss_workspace_bytes(a.B, A, nmax)
        if self._ws is None or self._ws.numel() < need:
            from ..hip.engine import dev_empty
            self._ws = dev_empty(need, torch.uint8, self.device)
        a.workspace = self._ws.data_ptr()
        self.A = A
        return a

    def set_targets(self, batch, cap=None):
        """Stage the (n,6)-style targets of the dataloader batch dict on the device (static buffers of capacity ``cap``)."""
        bi = batch["batch_idx"].reshape(-1).float()
        n = bi.numel()
        cap = max(cap or 0, n, 8)
        if self._tgt is None or self._tgt[0].numel() < cap:
            self._tgt = (torch.zeros(cap, device=self.device), torch.zeros(cap, device=self.device),
                         torch.zeros(cap, 4, device=self.device))
        if n:
            self._tgt[0][:n].copy_(bi, non_blocking=True)
            self._tgt[1][:n].copy_(batch["cls"].reshape(-1).float(), non_blocking=True)
            self._tgt[2][:n].copy_(batch["bboxes"].reshape(-1, 4).float(), non_blocking=True)
        a = self._args
        a.t_batch_idx, a.t_cls, a.t_boxes = (t.data_ptr() for t in self._tgt)
        a.n_targets = n
        if self._ncount is None:
            self._ncount = torch.zeros(1, dtype=torch.int32, device=self.device)
        # the count travels by value (a fill kernel's argument), not through a reused pinned word: the host may be several steps
        # ahead of the device, and a later step's count must not reach an earlier step's loss kernels
        self._ncount.fill_(n)
        a.n_targets_dev = self._ncount.data_ptr()
        return n

    def check_capacity(self, scalars_host=None):
        """Raise if any image since the last check carried more labels than the per-image capacity ``nmax`` the loss kernels
        were bound with (the packing kernel drops the surplus and raises the sticky flag scalars[10]).  The reference pads to
        ``counts.max()`` (utils/loss.py:337-343) and never drops a label.  Synchronises unless the scalars are passed in."""
        s = self.scalars.cpu() if scalars_host is None else scalars_host
        if float(s[10]) != 0.0:
            self.scalars[10] = 0.0
            raise RuntimeError(f"an image carried more than nmax={self._args.nmax} labels: the surplus ground truth was dropped. "
                               "Raise nmax (cfg key 'nmax', StepPlan(nmax=...)) to at least the largest per-image label count")

    def sync_modes(self):
        a, b = self._args, self.bbox_loss
        a.hyp_box, a.hyp_cls, a.hyp_dfl = self.box_gain, self.cls_gain, self.dfl_gain
        a.use_wiou, a.use_nwd, a.iou_ratio = int(b.use_wiseiou), int(b.nwd_loss), float(b.iou_ratio)

    @staticmethod
    def capacity_for(batch, B):
        bi = batch["batch_idx"].reshape(-1)
        if bi.numel() == 0:
            return 8
        counts = torch.bincount(bi.long().cpu(), minlength=B)
        return max(8, (int(counts.max()) + 7) // 8 * 8)

    def __call__(self, preds, batch):
        """(loss.sum()*B, loss_items[box, cls, dfl]) as device tensors (reference utils/loss.py:356-361).  ``preds``: the HIP
        path's HeadOut, or the reference's list of (B, no, H, W) maps / eval-mode (y, feats) tuple.  Runs on its own argument
        block, workspace and target buffers, so a call between two steps of a recorded StepPlan (which replays launches bound
        to THIS object's buffers) disturbs nothing; the running WIoU mean and the result scalars are shared, as in the
        reference where the criterion is one object."""
        from ..nn.modules.head import HeadOut
        if not isinstance(preds, HeadOut):
            preds = self._headout_from_reference(preds)
        preds.materialize()  # a forward that left its logits out (fused decode in a recorded step / the fused inference tail) writes them now
        pub = self.__dict__.get("_pub")
        if pub is None:
            pub = object.__new__(type(self))
            pub.__dict__.update({k: v for k, v in self.__dict__.items() if k != "_pub"})
            pub._args, pub._ws, pub._tgt, pub._ncount = DyLossArgs(), None, None, None
            self._pub = pub
        pub.box_gain, pub.cls_gain, pub.dfl_gain = self.box_gain, self.cls_gain, self.dfl_gain
        B = preds.box[0].shape[0]
        pub.bind(preds, self.capacity_for(batch, B))
        pub.set_targets(batch)
        pub.sync_modes()
        from ..hip import lib
        check(lib().dy_detection_loss(C.byref(pub._args), torch.cuda.current_stream(self.device).cuda_stream), "dy_detection_loss")
        self._last = pub
        return self.scalars[8].clone(), self.scalars[5:8].clone()

    def _headout_from_reference(self, preds):
        """The reference's ``preds`` protocol (utils/loss.py:356-368): the list of per-level ``(B, no, H, W)`` maps the head
        returns in training mode, or the ``(y, feats)`` tuple of eval mode (``feats = preds[1] if isinstance(preds, tuple)``).
        Re-laid as the kernels' fp32 NHWC (B,H,W,64) box logits + (B,H,W,ncp) class logits.  Loss values only."""
        from ..nn.modules.head import HeadOut, LazyFeats
        feats = preds[1] if isinstance(preds, tuple) else preds
        if isinstance(feats, LazyFeats):
            return feats._ho
        if not isinstance(feats, (list, tuple)) or not all(torch.is_tensor(f) and f.dim() == 4 and f.shape[1] == self.no for f in feats):
            raise TypeError(f"v8DetectionLoss expects the head's training output: a list of (B, {self.no}, H, W) tensors "
                            "(or the HeadOut of the HIP path)")
        nb, ncp = self.reg_max * 4, (self.nc + 7) // 8 * 8
        box, cls = [], []
        for f in feats:
            f = f.detach().to(self.device, torch.float32).permute(0, 2, 3, 1)
            box.append(f[..., :nb].contiguous())
            c = torch.zeros((*f.shape[:3], ncp), dtype=torch.float32, device=self.device)
            c[..., :self.nc] = f[..., nb:]
            cls.append(c)
        return HeadOut(box, cls, self.nc, [float(s) for s in self.stride])

    def assignment_rows(self):
        """(device pointer of the per-anchor assigned-gt index (B, A) int32 the loss kernels leave in the workspace, A, first anchor
        of each level) of the bound geometry: what dy_conv1x1_rows_backward reads instead of a dense box-logit gradient."""
        from ..hip import lib
        a = self._args
        o = [C.c_size_t() for _ in range(3)]
        lib().dy_loss_workspace_layout(a.B, self.A, a.nmax, *[C.byref(x) for x in o])
        a0, first = 0, []
        for l in range(a.nl):
            first.append(a0)
            a0 += a.H[l] * a.W[l]
        return self._ws.data_ptr() + o[1].value, self.A, first

    def pred_box_ptr(self):
        """Device pointer of the decoded boxes (B, A, 4) inside the bound workspace (what dy_head_box_decode fills)."""
        from ..hip import lib
        a = self._args
        o = [C.c_size_t() for _ in range(3)]
        lib().dy_loss_workspace_layout(a.B, self.A, a.nmax, *[C.byref(x) for x in o])
        return self._ws.data_ptr() + o[0].value

    def debug_assignment(self):
        """(target_gt_idx (B,A) with -1 for background, target score (B,A), pred boxes (B,A,4)) of the last call."""
        from ..hip import lib
        own = self.__dict__.get("_last", self)  # the public call works on its own buffers, a StepPlan on this object's
        a = own._args
        o = [C.c_size_t() for _ in range(3)]
        lib().dy_loss_workspace_layout(a.B, own.A, a.nmax, *[C.byref(x) for x in o])
        BA = a.B * own.A
        ws = own._ws
        pb = ws[o[0].value:o[0].value + BA * 16].view(torch.float32).view(a.B, own.A, 4)
        gt = ws[o[1].value:o[1].value + BA * 4].view(torch.int32).view(a.B, own.A)
        ts = ws[o[2].value:o[2].value + BA * 4].view(torch.float32).view(a.B, own.A)
        return gt.clone(), ts.clone(), pb.clone()
