"""Training loop of the hot path (drop-in for the step semantics of reference engine/trainer.py:595-957, 1115-1180):
warm-up interpolation of lr / momentum / accumulate, linear | cosine LR, parameter groups, global-norm clip 10, SGD-nesterov
| Adam(W) ('auto' rule), EMA, one process per GPU with a single RCCL all-reduce per step.  The batch source is any iterable
of the dataloader's batch dicts (img, batch_idx, cls, bboxes; reference data/dataset.py:207-224): the CPU data pipeline
(``ultralytics.data``: YOLO-format reader + pinned-buffer loader, augmentation-free) plugs in through ``get_dataloader`` /
``train(data=<yaml>)``."""
from __future__ import annotations

import math
import os
import random
import time

import numpy as np
import torch
import torch.distributed as dist

from ..cfg import get_cfg
from ..hip.train import StepPlan
from ..utils import LOGGER, RANK
from ..utils.torch_utils import ModelEMA, init_seeds, select_device


class DetectionTrainer:
    def __init__(self, model=None, cfg=None, overrides=None, _callbacks=None):
        """``DetectionTrainer(model, overrides=...)`` (this package's YOLO facade) or, as in the reference
        (engine/trainer.py:490-560), ``DetectionTrainer(cfg=..., overrides=dict(model=<yaml | pt>, data=<yaml>, ...)).train()``."""
        if model is not None and not isinstance(model, torch.nn.Module):  # reference call shape: first positional is cfg
            model, cfg = None, model
        self.args = get_cfg(cfg or {}, overrides) if cfg else get_cfg(overrides=overrides)
        if model is None:
            if not self.args.model:
                raise ValueError("DetectionTrainer needs a model: pass a DetectionModel or overrides['model'] = <model yaml | checkpoint>")
            from ..nn.tasks import DetectionModel, attempt_load_weights
            m = str(self.args.model)
            model = DetectionModel(m, verbose=False) if m.endswith((".yaml", ".yml")) else attempt_load_weights(m)
        self.model = model
        self.world_size = int(os.environ.get("WORLD_SIZE", 1))
        self.rank = int(os.environ.get("RANK", 0))
        self.rehearsal = os.environ.get("DY_REHEARSE_ON_ONE_GPU") == "1"  # N ranks sharing GPU 0 over gloo: exercises the DP path
        self.dataset_len = None
        init_seeds(self.args.seed + 1 + RANK, self.args.deterministic)  # reference engine/trainer.py:526 (RANK = -1 single process)
        self.device = torch.device("cuda", 0) if self.rehearsal else select_device(self.args.device)
        self.plan = self.ema = None
        self.lf = None

    # ---- reference build_optimizer 'auto' rule (engine/trainer.py:1133-1144)
    def _optimizer_choice(self, iterations, nc):
        name, lr, mom = self.args.optimizer, self.args.lr0, self.args.momentum
        if name == "auto":
            lr_fit = round(0.002 * 5 / (4 + nc), 6)
            name, lr, mom = ("SGD", 0.01, 0.9) if iterations > 10000 else ("AdamW", lr_fit, 0.9)
            self.args.warmup_bias_lr = 0.0
        if name not in ("SGD", "Adam", "AdamW", "RMSProp", "RAdam", "Adamax", "NAdam", "SOAP"):
            raise NotImplementedError(f"Optimizer '{name}' not found in list of available optimizers "
                                      "[Adam, AdamW, NAdam, RAdam, RMSProp, SGD, SOAP, auto].")
        return name, lr, mom

    def setup(self, batches_per_epoch, batch_size, imgsz):
        a = self.args
        if self.world_size > 1 and not dist.is_initialized():
            torch.cuda.set_device(self.device)
            if self.rehearsal:
                dist.init_process_group("gloo", rank=self.rank, world_size=self.world_size)
            else:
                dist.init_process_group("nccl", rank=self.rank, world_size=self.world_size, device_id=self.device)
        self.model.to(self.device).train()
        self.model.args = a
        if self.world_size > 1:
            # Every rank seeds with seed + 1 + RANK (reference engine/trainer.py:526) and a model built from a YAML after that differs
            # from rank to rank; the reference relies on DDP's constructor broadcasting rank 0's parameters and buffers
            # (engine/trainer.py:695).  Same here, on the flat buffers, BEFORE the plan clones them into its EMA.
            from ..hip.dist import broadcast_buffers
            rt = self.model._runtime(self.device)
            broadcast_buffers(rt.flat_p, 0)
            broadcast_buffers(rt.flat_b, 0)
            rt.mark_dirty()
        for k, v in self.model.named_parameters():  # always freeze .dfl (engine/trainer.py:670)
            v.requires_grad = ".dfl" not in k
        global_bs = batch_size * self.world_size
        self.accumulate = max(round(a.nbs / global_bs), 1)
        self.wd = a.weight_decay * global_bs * self.accumulate / a.nbs
        # reference engine/trainer.py:707: ceil(len(train_loader.dataset) / max(self.batch_size, nbs)) * epochs with the GLOBAL batch
        # and the whole dataset, whatever the number of ranks
        n_images = self.dataset_len if self.dataset_len is not None else batches_per_epoch * global_bs
        iterations = math.ceil(n_images / max(global_bs, a.nbs)) * a.epochs
        name, self.lr0, self.momentum = self._optimizer_choice(iterations, self.model.model[-1].nc)
        self.opt_name = name
        self._plan_kw = dict(nmax=getattr(a, "nmax", None) or 16, optimizer=name, world_size=self.world_size, use_graph=bool(a.hipgraph),
                             init_scale=float(a.loss_scale) if a.amp else 1.0, dynamic_scale=bool(a.amp))
        self.plan = StepPlan(self.model, batch_size, imgsz, **self._plan_kw)
        self.plans = {batch_size: self.plan}  # forward/backward launch lists by batch size (the ragged last batch gets its own)
        self.arena = None  # multi_scale: one block of HBM under the step-local buffers of every size's launch list (_scaled_plan)
        self.gs = max(int(self.model.stride.max()), 32)  # reference engine/trainer.py:677: grid size, never below 32
        bl = self.plan.crit.bbox_loss
        bl.use_wiseiou, bl.nwd_loss, bl.iou_ratio = bool(a.wiou), bool(a.nwd), float(a.iou_ratio)
        self.ema = ModelEMA(self.plan)
        self.nb = batches_per_epoch
        self.nw = max(round(a.warmup_epochs * self.nb), 100) if a.warmup_epochs > 0 else -1
        if a.cos_lr:
            self.lf = lambda x: max((1 - math.cos(x * math.pi / a.epochs)) / 2, 0) * (a.lrf - 1) + 1
        else:
            self.lf = lambda x: max(1 - x / a.epochs, 0) * (1.0 - a.lrf) + a.lrf
        self.last_opt_step = -1

    def train_step(self, batch, ni, epoch):
        """One iteration of the hot loop (reference engine/trainer.py:780-815)."""
        a, p = self.args, self.plan
        lr = [self.lr0 * self.lf(epoch)] * 3
        mom = self.momentum
        acc = self.accumulate
        if ni <= self.nw:
            xi = [0, self.nw]
            acc = max(1, int(np.interp(ni, xi, [1, a.nbs / (p.B * self.world_size)]).round()))
            lr = [float(np.interp(ni, xi, [a.warmup_bias_lr if j == 0 else 0.0, self.lr0 * self.lf(epoch)])) for j in range(3)]
            if self.opt_name in ("SGD", "RMSProp"):  # reference :791 ``if "momentum" in x``: only these param groups have the key;
                mom = float(np.interp(ni, xi, [a.warmup_momentum, self.momentum]))  # Adam-family / SOAP betas are never warmed up
        p.set_hyper(lr, mom, [0.0, self.wd, 0.0])
        # (gradient buckets: the early all-reduce sums the gradient buffer in place, so it may only start when this micro-batch is
        # the whole optimizer step)
        alone = not (self.accumulate > 1 or acc > 1 or p._micro) and ni - self.last_opt_step >= acc
        if a.multi_scale:
            self._forward_backward_rescaled(batch, exchange=alone)
        else:
            self._plan_for(batch).forward_backward(batch, exchange=alone)  # writes the shared flat gradient buffer; everything below is the main plan's
        if self.accumulate > 1 or acc > 1 or p._micro:
            p.accumulate()
        if ni - self.last_opt_step >= acc:
            p.all_reduce()
            p.optimizer_step()
            self.last_opt_step = ni
        return lr, mom

    def _plan_for(self, batch):
        """The recorded forward/backward for this batch's size: the reference trains on the ragged last batch of an epoch too
        (data/build.py:104-124 has no drop_last), a recorded launch list has one batch size -> one plan per size, optimizer
        state shared (StepPlan(share=...))."""
        img = batch["img"]
        B = len(batch["index"]) if "index" in batch else (len(batch["warp"]) if "warp" in batch else img.shape[0])
        fb = self.plans.get(B)
        if fb is None:
            fb = self.plans[B] = StepPlan(self.model, B, self.plan.imgsz, share=self.plan, **self._plan_kw)
        return fb

    # ---- multi_scale (reference models/yolo/detect/train.py:60-73) ---------------------------------------------------------------
    def _forward_backward_rescaled(self, batch, exchange=True):
        """``preprocess_batch`` with ``multi_scale=True``: every batch is bilinearly re-interpolated to a random multiple of the grid
        size in [0.5, 1.5] x imgsz before the step.  A recorded launch list has one input size, so each size gets its own list
        (``input_act`` plans fed through ``x_in``); they all lay their step-local buffers over ONE arena sized for the largest size --
        the footprint of a multi-scale run is that of its largest step, not the sum over sizes -- and share the optimizer state."""
        a, gs = self.args, self.gs
        sz = random.randrange(int(a.imgsz * 0.5), int(a.imgsz * 1.5 + gs)) // gs * gs
        base = self._plan_for(batch)  # the plan of this batch size at the base size: stages the loader's batch (pixels, flips, HSV, pool)
        H, W = base.imgsz
        sf = sz / max(H, W)
        if sf == 1:
            return base.forward_backward(batch, exchange=exchange)
        ns = tuple(math.ceil(x * sf / gs) * gs for x in (H, W))
        base.stage(batch)
        x = base.import_input()  # eager launch: fp16 NHWC, 8 channels (3 used)
        if not hasattr(x, "st"):  # fp32 NCHW batches travel as an ImageAct (direct stem): here the imported copy is what is resized
            x = x.materialize()
        plan = self._scaled_plan(base, ns)
        img = torch.nn.functional.interpolate(x.st.buf[..., :3].permute(0, 3, 1, 2).float(), size=ns, mode="bilinear", align_corners=False)
        plan.x_in[..., :3].copy_(img.permute(0, 2, 3, 1))
        return plan.forward_backward(batch, exchange=exchange)

    def _scaled_plan(self, base, ns):
        from ..hip.engine import Arena
        key = (base.B, *ns)
        plan = self.plans.get(key)
        if plan is not None:
            return plan
        kw = dict(self._plan_kw, share=self.plan, input_act=True)
        if self.arena is None:  # sized once, on the largest size this run can draw, with the main batch size
            a, gs = self.args, self.gs
            top = (int(a.imgsz * 1.5 + gs) - 1) // gs * gs
            big = tuple(math.ceil(x * (top / max(self.plan.imgsz)) / gs) * gs for x in self.plan.imgsz)
            probe_arena = Arena(0, self.device, measure=True)
            eng, kept = self.plan.eng, len(self.plan.eng.keep)
            probe = StepPlan(self.model, self.plan.B, big, arena=probe_arena, **dict(kw, use_graph=False))
            saved = self.plan.rt.flat_b.clone(), self.plan.crit.scalars.clone()  # the probe step must leave no trace: BatchNorm
            n0 = self.plan.B  # running statistics and the loss scalars (WIoU running mean) back; gradients are rewritten by the next step
            dummy = dict(batch_idx=torch.arange(n0, dtype=torch.float32), cls=torch.zeros((n0, 1)), bboxes=torch.full((n0, 4), 0.5))
            probe.forward_backward(dummy)
            torch.cuda.synchronize()
            self.plan.rt.flat_b.copy_(saved[0])
            self.plan.crit.scalars.copy_(saved[1])
            need = probe_arena.peak
            del probe, probe_arena, eng.keep[kept:]  # the engine keeps recorded buffers alive: the probe's list is gone, so are they
            torch.cuda.empty_cache()
            self.arena = Arena(need + (64 << 20), self.device)
            LOGGER.info(f"multi_scale: {self.arena.cap / 2**30:.2f} GiB arena shared by every input size (largest {big[0]}x{big[1]})")
        plan = self.plans[key] = StepPlan(self.model, base.B, ns, arena=self.arena if self.share_arena else None, **kw)
        return plan

    share_arena = True  # False: every size keeps its own step-local buffers (what the arena is tested against)

    def train(self, loader=None, batch_size=None, imgsz=None, epochs=None, log_every=0):
        """loader: re-iterable of batch dicts (``batch_size`` = its per-rank batch).  Without arguments -- the reference's call shape
        -- trains on ``args.data`` with ``args.batch`` / ``args.imgsz``.  Returns the list of per-epoch mean loss items."""
        a = self.args
        if loader is None:
            if not a.data:
                raise ValueError("train() without a loader needs overrides['data'] = <dataset yaml>")
            return self.train_on_dataset(a.data, a.batch, a.imgsz, log_every=log_every)
        if epochs is not None:
            a.epochs = epochs
        nb = len(loader)
        if not getattr(a, "nmax", None):  # per-image label capacity of the recorded loss kernels; exceeded later -> raises
            cap = self._label_capacity(loader)
            if cap is None:  # an opaque iterable: take it from the first batch, with head-room.  (A dataset-backed loader is NOT
                from ..utils.loss import v8DetectionLoss  # peeked: starting its producer would consume augmentation draws.)
                first = next(iter(loader))
                cap = max(16, 2 * v8DetectionLoss.capacity_for(first, batch_size))
                del first
            a.nmax = cap
        self.setup(nb, batch_size, imgsz)
        self.start_epoch = self.resume_training(a.resume) if a.resume else 0
        self.save_dir = self._save_dir()
        hist = []
        for epoch in range(self.start_epoch, a.epochs):
            if getattr(self, "_epoch_hook", None):
                self._epoch_hook(epoch)  # DistributedSampler.set_epoch (engine/trainer.py:766-767)
            ds = getattr(loader, "dataset", None)
            if ds is not None and getattr(ds, "mosaic", 0.0) and epoch == a.epochs - a.close_mosaic:
                LOGGER.info("Closing dataloader mosaic")  # _close_dataloader_mosaic (engine/trainer.py:772-776, :934-940)
                ds.mosaic = 0.0  # the affine / HSV / flip draws go on; batches keep their warp records
            t0, tloss = time.time(), None
            for i, batch in enumerate(loader):
                ni = i + nb * epoch
                self.train_step(batch, ni, epoch)
                if ni % 256 == 255:  # long epochs: do not wait for the epoch end to notice steps that do not take effect
                    self.plan.check_progress()
                if log_every and (i % log_every == 0):
                    _, items = self.plan.loss_items()
                    tloss = items if tloss is None else (tloss * i + items) / (i + 1)
            torch.cuda.synchronize()
            _, items = self.plan.loss_items()
            self.plan.check_progress()  # raises when optimizer steps are not taking effect (counter stuck, all skipped, scale collapsed)
            hist.append(items if tloss is None else tloss)
            self.epoch = epoch
            self._check_replicas()
            if self.rank == 0:
                LOGGER.info(f"epoch {epoch + 1}/{a.epochs}  box/cls/dfl {[round(float(x), 4) for x in hist[-1]]}  "
                            f"{nb * batch_size * self.world_size / (time.time() - t0):.1f} img/s")
                if self.save_dir is not None:  # reference engine/trainer.py:898-923: last.pt after every epoch
                    self.save_model(self.save_dir / "weights" / "last.pt")
        return hist

    def _check_replicas(self):
        """Data-parallel replicas must hold identical weights after every optimizer step (same start: setup()'s broadcast; same
        summed gradients: the step's one all-reduce).  One checksum exchange per epoch makes a divergence an error instead of a
        silently worse model."""
        if self.world_size <= 1 or not dist.is_initialized():
            return
        rt = self.plan.rt
        mine = torch.stack([rt.flat_p.double().sum(), rt.flat_p.double().abs().sum()])
        if dist.get_backend() == "gloo":  # rehearsal: gloo moves host memory
            mine = mine.cpu()
        allv = [torch.zeros_like(mine) for _ in range(self.world_size)]
        dist.all_gather(allv, mine)
        allv = [v.cpu() for v in allv]
        if any(not torch.equal(v, allv[0]) for v in allv[1:]):
            raise RuntimeError(f"data-parallel replicas diverged (parameter checksums per rank: {[v.tolist() for v in allv]})")

    def _label_capacity(self, loader):
        """Largest number of boxes one training sample can carry, from the dataset's label table: a mosaic holds four images'
        boxes, MixUp concatenates two samples (reference data/augment.py: Mosaic / MixUp), so up to eight images' boxes meet in
        one sample.  The mosaic partners are drawn WITH replacement from a buffer that already holds the index image
        (data/dataset.py: ``random.choices(list(self.buffer), k=3)``) and MixUp's partner is an independent draw, so the image with
        the most labels can fill every tile: the bound is tiles x the largest count, not the sum of the largest distinct counts.
        None for loaders without a label table."""
        ds = getattr(loader, "dataset", None)
        labels = getattr(ds, "labels", None)
        if not labels:
            return None
        most = max((len(lb["cls"]) for lb in labels), default=1)
        per = 4 if getattr(ds, "mosaic", 0.0) else 1
        if getattr(self.args, "mixup", 0.0):
            per *= 2
        return max(8, (per * most + 7) // 8 * 8)

    def _save_dir(self):
        """``project/name`` as in the reference (cfg get_save_dir), created only when one of the two was given explicitly: this
        package does not write run folders unasked (the reference always creates runs/detect/trainN)."""
        from pathlib import Path
        a = self.args
        if not (a.save and (a.project or a.name)) or self.rank != 0:
            return None
        d = Path(a.project or "runs/detect") / (a.name or "train")
        (d / "weights").mkdir(parents=True, exist_ok=True)
        return d

    def resume_training(self, resume):
        """reference engine/trainer.py:1050-1105 ``check_resume`` + ``resume_training``: restore the optimizer state (momentum / Adam
        moments, loss scale and step counters), the EMA and its update count and the weights from a checkpoint written by
        ``save_model`` (this package's format), and continue the schedule at the epoch after the saved one."""
        path = resume if isinstance(resume, str) else self.args.model
        if not path or not str(path).endswith(".pt"):
            raise FileNotFoundError("resume=True needs the checkpoint: YOLO('<run>/weights/last.pt').train(resume=True) or resume='<path>'")
        ck = torch.load(path, map_location="cpu", weights_only=False)
        opt = ck.get("optimizer") if isinstance(ck, dict) else None
        if not isinstance(opt, dict) or "flat" not in opt:
            raise ValueError(f"{path} carries no resumable optimizer state (written by DetectionTrainer.save_model of this package?)")
        p, rt = self.plan, self.plan.rt
        if sorted(opt["param_names"]) != sorted(rt.param_names) or opt["mode"] != p.mode:
            raise ValueError("the checkpoint's parameters / optimizer do not match this model and optimizer")
        f = dict(opt["flat"])
        if list(opt["param_names"]) != list(rt.param_names):
            # a checkpoint written under another ORDER of the flat parameter buffer (before round 4: [bias | decayed | norm] over the whole
            # model; now the neck + head in front of the backbone, hip/runtime.py): move every tensor's slice to where it lives now
            sizes = {n: prm.numel() for n, prm in self.model.named_parameters()}
            old_off, o = {}, 0
            for n in opt["param_names"]:
                old_off[n] = o
                o += (sizes[n] + 7) // 8 * 8
            for key in ("p", "ema", "mom", "adam_v"):
                if key in f:
                    src, dst = f[key], torch.zeros(rt.n_params_flat, dtype=f[key].dtype)
                    for n, k in sizes.items():
                        dst[rt.param_off[n]:rt.param_off[n] + k] = src[old_off[n]:old_off[n] + k]
                    f[key] = dst
        for dst, key in ((rt.flat_p, "p"), (rt.flat_b, "b"), (p.ema, "ema"), (p.ema_b, "ema_b"), (p.mom, "mom"), (p.state, "state")):
            dst.copy_(f[key])
        if p.adam_v is not None:
            p.adam_v.copy_(f["adam_v"])
        if p.soap:
            if "soap" not in opt:
                raise ValueError(f"{path} was written without the SOAP preconditioner state: a resume would silently restart the optimizer")
            p.load_soap_state(opt["soap"])
        p.ema_updates, self.last_opt_step = int(ck["updates"]), int(opt["last_opt_step"])
        p.opt_calls = int(f["state"][5]) + int(f["state"][6])
        rt.mark_dirty()
        LOGGER.info(f"Resuming training from {path} from epoch {int(ck['epoch']) + 2} to {self.args.epochs} total epochs")
        return int(ck["epoch"]) + 1

    # ---- dataset-backed entry points (reference models/yolo/detect/train.py:33-55, engine/trainer.py:517-548) -------------
    def get_dataloader(self, dataset_path, batch_size=16, rank=0, mode="train", data=None):
        """Loader over a YOLO-format image folder.  Train: shuffled batches, the ragged last one included as in the reference (it
        runs through its own recorded launch list, ``_plan_for``), images delivered as uint8 NHWC on the device;
        val: the reference's rectangular batches, uint8 NCHW on the host side until ``preprocess``."""
        from ..data import build_dataloader, build_yolo_dataset
        a = self.args
        if mode == "train":
            # HSV jitter and flips run in the import kernels; mosaic / affine / perspective / MixUp pixels are composed on the device
            # from the HBM image pool; copy_paste is a no-op for box-only labels, as in the reference
            geo = [k for k in ("mosaic", "degrees", "translate", "scale", "shear", "perspective", "mixup") if getattr(a, k, 0)]
            if geo and a.cache != "hbm":
                LOGGER.warning(f"WARNING augmentations {geo} need cache='hbm' (device-side composition): training without them")
                a = type(a)(**{**vars(a), **{k: 0.0 for k in geo}})
        ds = build_yolo_dataset(a, dataset_path, batch_size, data, mode=mode, rect=mode == "val", stride=32,
                                layout="nhwc" if mode == "train" else "nchw", flip_on_device=mode == "train")
        return build_dataloader(ds, batch_size, a.workers, shuffle=mode == "train", rank=rank if self.world_size > 1 else -1,
                                world_size=self.world_size, device=self.device if mode == "train" else None, drop_last=False)

    def train_on_dataset(self, data_yaml, batch_size, imgsz, log_every=0):
        """``YOLO.train(data=<yaml>)``: check the dataset YAML, build the loaders, train, validate the EMA model on 'val'."""
        from ..data import check_det_dataset
        from ..models.yolo.detect import DetectionValidator
        self.data = check_det_dataset(data_yaml)
        self.model.names = self.data["names"]
        if self.world_size > 1:  # ``batch`` is the GLOBAL batch, split over the ranks (reference engine/trainer.py:692)
            batch_size = max(batch_size // self.world_size, 1)
        loader = self.get_dataloader(self.data["train"], batch_size, self.rank, "train", self.data)
        self.dataset_len = len(loader.dataset)
        batch_size = loader.batch_size  # build_dataloader clamps it to the dataset size, as the reference does (data/build.py:104)
        if len(loader) == 0:
            raise ValueError(f"the training split holds {len(loader.dataset)} images, fewer than one batch of {batch_size}")
        self._epoch_hook = loader.set_epoch
        self.train_loader = loader
        # per-image label capacity of the recorded loss kernels: derived from the label table, never below what the user asked for
        self.args.nmax = max(int(getattr(self.args, "nmax", 0) or 0), self._label_capacity(loader) or 8)
        hist = self.train(loader, batch_size, imgsz, log_every=log_every)  # log_every=1: per-epoch MEAN loss items, as results.csv
        self.metrics = None
        if self.args.val and self.rank == 0:
            vloader = self.get_dataloader(self.data["val"], batch_size * 2, 0, "val", self.data)
            self.validator = DetectionValidator(dataloader=vloader, args=self.args)
            self.metrics = self.validator(model=self.ema.ema)
        return hist

    def save_model(self, path, reference_format=False):
        """state_dict checkpoint (model + EMA), or -- ``reference_format=True`` -- the reference's whole-module fp16 layout
        (engine/trainer.py:898-923) that the reference's ``attempt_load_weights`` reads directly."""
        if reference_format:
            from ..nn.tasks import save_reference_format
            return save_reference_format(path, self.model, ema=self.ema.ema, updates=self.plan.ema_updates, train_args=vars(self.args))
        p, rt = self.plan, self.plan.rt
        flat = {"p": rt.flat_p, "b": rt.flat_b, "ema": p.ema, "ema_b": p.ema_b, "mom": p.mom, "state": p.state}
        if p.adam_v is not None:
            flat["adam_v"] = p.adam_v
        torch.save({"epoch": getattr(self, "epoch", -1),
                    "model": {k: v.detach().cpu() for k, v in self.model.state_dict().items()},
                    "ema": {k: v.detach().cpu() for k, v in self.ema.state_dict().items()},
                    "updates": p.ema_updates, "train_args": vars(self.args), "yaml": self.model.yaml,
                    # what resume needs (reference engine/trainer.py:911 keeps optimizer.state_dict()): the flat optimizer buffers
                    "optimizer": {"mode": p.mode, "param_names": list(rt.param_names), "last_opt_step": self.last_opt_step,
                                  "flat": {k: v.detach().cpu().clone() for k, v in flat.items()},
                                  **({"soap": p.soap_state()} if p.soap else {})}}, path)
