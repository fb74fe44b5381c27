"""``YOLO`` facade (drop-in for the detect task of reference engine/model.py + models/yolo/model.py)."""
from __future__ import annotations

import json
import os
import subprocess
import tempfile
from pathlib import Path

import torch

from ..cfg import DEFAULT_CFG_DICT
from ..nn.tasks import DetectionModel, attempt_load_weights
from ..utils import LOGGER, RANK
from .trainer import DetectionTrainer


class YOLO:
    def __init__(self, model="yolov8n-ASF-P2P2.yaml", task="detect", verbose=False):
        if task not in (None, "detect"):
            raise NotImplementedError("only the detect task is on the DEAL-YOLO hot path")
        self.task, self.ckpt_path, self.cfg = "detect", None, None
        self.model_name = str(model)
        if Path(str(model)).suffix in (".yaml", ".yml"):
            self.model = DetectionModel(model, verbose=verbose)
            self.cfg = str(model)
        else:
            self.model = attempt_load_weights(model)
            self.ckpt_path = str(model)
        self.trainer = self.predictor = self.validator = None
        self.overrides = {}

    # ---- reference engine/model.py surface -----------------------------------------------------------------------
    @property
    def task_map(self):
        """reference models/yolo/model.py:21-26 (detect entry)."""
        from ..models.yolo.detect import DetectionPredictor, DetectionTrainer as T, DetectionValidator
        return {"detect": {"model": DetectionModel, "trainer": T, "validator": DetectionValidator, "predictor": DetectionPredictor}}

    @property
    def names(self):
        return self.model.names

    @property
    def device(self):
        return next(self.model.parameters()).device

    def fuse(self):
        self.model.fuse()
        return self

    def info(self, detailed=False, verbose=True):
        return self.model.info(detailed, verbose)

    def load(self, weights):
        ck = torch.load(weights, map_location="cpu", weights_only=False) if isinstance(weights, (str, Path)) else weights
        self.model.load(ck)
        return self

    def to(self, device):
        self.model.to(device)
        return self

    # ---- train ---------------------------------------------------------------------------------------------------
    def train(self, trainer=None, **kwargs):
        """reference engine/model.py:548-620 ``model.train(data=..., epochs=..., batch=..., device=..., ...)``.

        ``data``: a dataset YAML (YOLO-format folders; reference data/utils.py:252) or -- an addition -- a re-iterable of batch
        dicts (img float [0,1] | uint8, batch_idx, cls, bboxes), e.g. ``ultralytics.data.SyntheticDetection``.
        ``device='0,1,...'``: one rank per listed GPU, re-launched under torch.distributed.run before any GPU call."""
        data = kwargs.pop("data", None)
        if data is None:
            raise ValueError("data=<dataset yaml> or an iterable of batch dicts is required")
        log_every = kwargs.pop("log_every", 0)
        batch = kwargs.get("batch", DEFAULT_CFG_DICT["batch"])
        imgsz = kwargs.get("imgsz", DEFAULT_CFG_DICT["imgsz"])
        from ..utils.dist import parse_devices
        devices = parse_devices(kwargs.get("device", DEFAULT_CFG_DICT["device"]))
        if len(devices) > 1 and RANK == -1 and "WORLD_SIZE" not in os.environ:
            if not isinstance(data, (str, Path)):
                raise ValueError("multi-GPU training re-launches itself: data must be a dataset YAML path, not an in-memory iterable")
            return self._train_ddp(devices, dict(kwargs, data=str(data)))
        T = trainer or DetectionTrainer
        if kwargs.get("resume") and self.ckpt_path and kwargs.get("resume") is True:
            kwargs["resume"] = self.ckpt_path  # YOLO('<run>/weights/last.pt').train(resume=True), reference engine/model.py:579-581
        if isinstance(data, (str, Path)):
            from ..data import check_det_dataset
            nc = check_det_dataset(data)["nc"]
            if self.ckpt_path is not None and nc != self.model.model[-1].nc:
                raise ValueError(f"checkpoint has {self.model.model[-1].nc} classes, dataset has {nc}")
            self.trainer = T(self.model, overrides=dict(kwargs, data=str(data)))  # seeds the RNGs: init_seeds(seed + 1 + RANK)
            self._fresh_model(nc)
            return self.trainer.train_on_dataset(data, batch, imgsz, log_every=log_every)
        self.trainer = T(self.model, overrides=dict(kwargs))
        self._fresh_model(self.model.model[-1].nc)
        return self.trainer.train(data, batch, imgsz)

    def _fresh_model(self, nc):
        """reference engine/model.py:583-586: ``trainer.model = trainer.get_model(weights=self.model if self.ckpt else None,
        cfg=self.model.yaml)`` -- training from a YAML always starts from a model built AFTER the trainer seeded the RNGs, with the
        dataset's class count (detect/train.py:75-81), so ``seed`` alone decides the initial weights, as in the reference."""
        if self.ckpt_path is None:
            self.model = self.trainer.model = DetectionModel(self.model.yaml, nc=nc, verbose=False)

    def _train_ddp(self, devices, overrides):
        """reference engine/trainer.py:607-627: spawn ``torch.distributed.run`` with one rank per GPU and wait for it."""
        from ..utils.dist import ddp_cleanup, generate_ddp_command
        model_path = self.ckpt_path or self.model.yaml.get("yaml_file") or self.cfg
        if not model_path:
            raise ValueError("multi-GPU training needs the model to come from a YAML file or a checkpoint path (the ranks rebuild it)")
        env = dict(os.environ)
        if env.get("DY_REHEARSE_ON_ONE_GPU") != "1":  # rehearsal: the ranks share GPU 0 and talk over gloo (tests on a one-GPU box)
            env["CUDA_VISIBLE_DEVICES"] = env["HIP_VISIBLE_DEVICES"] = ",".join(str(d) for d in devices)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        cmd, file, result_file = generate_ddp_command(len(devices), str(model_path), overrides)
        try:
            LOGGER.info(f"DDP: debug command {' '.join(cmd)}")
            subprocess.run(cmd, check=True, env=env)
            self.ddp_result = json.load(open(result_file)) if os.path.exists(result_file) else None
        finally:
            ddp_cleanup(file)
        # reference engine/model.py:612-616: after a DDP run the parent's model is the one the ranks trained (rank 0's checkpoint),
        # never the untouched copy this process was started with
        w = None if self.ddp_result is None else self.ddp_result.get("weights")
        if not w or not os.path.exists(w):
            raise RuntimeError("the multi-GPU run returned no weights (rank 0 writes <run>/weights/last.pt or a temporary checkpoint)")
        ck = torch.load(w, map_location="cpu", weights_only=False)
        names = getattr(self.model, "names", None)
        # the ranks' model: the dataset's class count and -- as attempt_load_one_weight does there (``ckpt.get("ema") or ckpt["model"]``,
        # reference nn/tasks.py:780) and attempt_load_weights does here -- the EMA weights when the checkpoint carries them, so that
        # val() / predict() after train(device='0,1') score the same weights as YOLO(last.pt) would
        self.model = DetectionModel(ck["yaml"], verbose=False)
        ema = ck.get("ema")
        self.model.load_state_dict(ema if isinstance(ema, dict) and ema else ck["model"], strict=True)
        if names is not None and len(names) == len(self.model.names):
            self.model.names = names
        tmp = w.startswith(os.path.join(tempfile.gettempdir(), "dealyolo_ddp"))
        self.ckpt_path = None if tmp else w
        if tmp:  # a hand-over file, not a run artefact
            os.remove(w)
        return self.ddp_result["loss_items"]

    # ---- predict -------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def predict(self, source=None, stream=False, predictor=None, **kwargs):
        """reference engine/model.py:386-440 ``model.predict(source=..., imgsz=..., conf=..., ...)`` -> list of ``Results``.
        ``source``: image file | directory | glob | list | PIL | BGR ndarray | (B,3,H,W) tensor in [0,1]."""
        from .predictor import DetectionPredictor
        args = {**self.overrides, "conf": 0.25, "batch": 1, **kwargs, "mode": "predict"}
        verbose = args.pop("verbose", False)  # the reference's default is True (it logs every image)
        if self.predictor is None or predictor is not None:
            self.predictor = (predictor or DetectionPredictor)(overrides=dict(args, verbose=verbose))
        else:
            from ..cfg import get_cfg
            self.predictor.args = get_cfg(vars(self.predictor.args), dict(args, verbose=verbose))
        self.predictor.setup_model(self.model, pt=self.ckpt_path is not None)
        return self.predictor(source=source, stream=stream)

    def __call__(self, source=None, stream=False, **kwargs):
        return self.predict(source, stream, **kwargs)

    # ---- val -----------------------------------------------------------------------------------------------------
    def val(self, validator=None, **kwargs):
        """``data``: a dataset YAML (its ``split``, rectangular batches as in the reference) or a re-iterable of batch dicts.
        Returns the metrics dict of the reference's DetMetrics.results_dict (precision, recall, mAP50, mAP50-95, fitness)."""
        from ..models.yolo.detect import DetectionValidator
        data, batch, split = kwargs.pop("data", None), kwargs.pop("batch", 32), kwargs.pop("split", "val")
        if data is None:
            raise ValueError("data=<dataset yaml> or an iterable of batch dicts is required")
        V = validator or DetectionValidator
        if isinstance(data, (str, Path)):
            from ..data import check_det_dataset
            d = check_det_dataset(data)
            tr = DetectionTrainer(self.model, overrides=dict(batch=batch, **kwargs))
            self.model.names = d["names"]
            if not d.get(split):
                raise KeyError(f"the dataset YAML has no '{split}' split")
            data = tr.get_dataloader(d[split], batch, 0, "val", d)
            self.validator = V(dataloader=data, args=tr.args)
        else:
            self.validator = V(dataloader=data, args=kwargs or None)
        return self.validator(model=self.model)
