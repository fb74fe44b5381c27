"""``YOLO`` facade (drop-in for the detect task of reference engine/model.py + models/yolo/model.py)."""
from __future__ import annotations

from pathlib import Path

import torch

from ..nn.tasks import DetectionModel, attempt_load_weights
from ..utils import ops
from .trainer import DetectionTrainer


class YOLO:
    def __init__(self, model="yolov8n-ASF-P2P2.yaml", task="detect", verbose=False):
        if task != "detect":
            raise NotImplementedError("only the detect task is on the DEAL-YOLO hot path")
        self.task, self.ckpt_path = task, None
        if Path(str(model)).suffix in (".yaml", ".yml"):
            self.model = DetectionModel(model, verbose=verbose)
        else:
            self.model = attempt_load_weights(model)
            self.ckpt_path = model
        self.trainer = None

    def fuse(self):
        self.model.fuse()
        return self

    def info(self, detailed=False, verbose=True):
        return self.model.info(detailed, verbose)

    def load(self, weights):
        ck = torch.load(weights, map_location="cpu", weights_only=False) if isinstance(weights, (str, Path)) else weights
        self.model.load(ck)
        return self

    def train(self, data=None, batch=16, imgsz=640, **kw):
        """``data``: a dataset YAML (YOLO-format folders; reference data/utils.py:252) or a re-iterable of batch dicts (img float
        [0,1] | uint8, batch_idx, cls, bboxes), e.g. ultralytics.data.SyntheticDetection(...)."""
        if data is None:
            raise ValueError("data=<dataset yaml> or an iterable of batch dicts is required")
        if isinstance(data, (str, Path)):
            from ..data import check_det_dataset
            nc = check_det_dataset(data)["nc"]
            if nc != self.model.model[-1].nc:  # the reference rebuilds the model with the dataset's class count
                if self.ckpt_path is not None:
                    raise ValueError(f"checkpoint has {self.model.model[-1].nc} classes, dataset has {nc}")
                self.model = DetectionModel(self.model.yaml, nc=nc, verbose=False)
            log_every = kw.pop("log_every", 0)
            self.trainer = DetectionTrainer(self.model, overrides=dict(batch=batch, imgsz=imgsz, data=str(data), **kw))
            return self.trainer.train_on_dataset(data, batch, imgsz, log_every=log_every)
        self.trainer = DetectionTrainer(self.model, overrides=dict(batch=batch, imgsz=imgsz, **kw))
        return self.trainer.train(data, batch, imgsz)

    @torch.no_grad()
    def predict(self, source, conf=0.25, iou=0.7, max_det=300, classes=None, agnostic_nms=False, device="0", **kw):
        """source: (N,3,H,W) float tensor in [0,1].  Returns the NMS output list (reference detect/predict.py:23-43)."""
        from ..utils.torch_utils import select_device
        dev = select_device(device)
        self.model.to(dev).eval()
        y, _ = self.model(source.to(dev))
        return ops.non_max_suppression(y, conf, iou, classes=classes, agnostic=agnostic_nms, max_det=max_det)

    __call__ = predict

    def val(self, data=None, batch=32, split="val", **kw):
        """``data``: a dataset YAML (its ``split``, rectangular batches as in the reference) or a re-iterable of batch dicts.
        Returns the metrics dict of the reference's DetMetrics.results_dict (precision, recall, mAP50, mAP50-95, fitness)."""
        from ..models.yolo.detect import DetectionValidator
        if data is None:
            raise ValueError("data=<dataset yaml> or an iterable of batch dicts is required")
        if isinstance(data, (str, Path)):
            from ..data import check_det_dataset
            d = check_det_dataset(data)
            tr = DetectionTrainer(self.model, overrides=dict(batch=batch, **kw))
            self.model.names = d["names"]
            if not d.get(split):
                raise KeyError(f"the dataset YAML has no '{split}' split")
            data = tr.get_dataloader(d[split], batch, 0, "val", d)
            self.validator = DetectionValidator(dataloader=data, args=tr.args)
        else:
            self.validator = DetectionValidator(dataloader=data, args=kw or None)
        return self.validator(model=self.model)
