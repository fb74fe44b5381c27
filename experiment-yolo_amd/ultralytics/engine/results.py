"""Detection results containers (drop-in for the parts of reference engine/results.py the detect task returns: ``Results`` with
``.boxes`` (``Boxes``: xyxy / conf / cls / xywh / xyxyn / xywhn / data), ``orig_img``, ``orig_shape``, ``path``, ``names``).
Drawing, saving and the other tasks' fields (masks, keypoints, probs, obb) are control plane / other tasks and absent."""
from __future__ import annotations

import numpy as np
import torch

from ..utils import ops


class Boxes:
    """(n, 6) detections [x1, y1, x2, y2, conf, cls] in ORIGINAL-image pixels (reference engine/results.py:405-470)."""

    def __init__(self, boxes, orig_shape):
        if boxes.ndim == 1:
            boxes = boxes[None, :]
        assert boxes.shape[-1] == 6, f"expected 6 values per box, got {boxes.shape[-1]}"
        self.data, self.orig_shape = boxes, tuple(orig_shape)
        self.is_track = False

    xyxy = property(lambda s: s.data[:, :4])
    conf = property(lambda s: s.data[:, -2])
    cls = property(lambda s: s.data[:, -1])
    id = property(lambda s: None)
    shape = property(lambda s: s.data.shape)

    @property
    def xywh(self):
        return ops.xyxy2xywh(self.xyxy)

    @property
    def xyxyn(self):
        xyxy = self.xyxy.clone() if isinstance(self.xyxy, torch.Tensor) else np.copy(self.xyxy)
        xyxy[..., [0, 2]] /= self.orig_shape[1]
        xyxy[..., [1, 3]] /= self.orig_shape[0]
        return xyxy

    @property
    def xywhn(self):
        xywh = ops.xyxy2xywh(self.xyxy)
        xywh[..., [0, 2]] /= self.orig_shape[1]
        xywh[..., [1, 3]] /= self.orig_shape[0]
        return xywh

    def __len__(self):
        return len(self.data)

    def __getitem__(self, idx):
        return Boxes(self.data[idx], self.orig_shape)

    def cpu(self):
        return Boxes(self.data.cpu(), self.orig_shape) if isinstance(self.data, torch.Tensor) else self

    def numpy(self):
        return Boxes(self.data.cpu().numpy(), self.orig_shape) if isinstance(self.data, torch.Tensor) else self

    def cuda(self):
        return Boxes(torch.as_tensor(self.data).cuda(), self.orig_shape)

    def to(self, *a, **k):
        return Boxes(torch.as_tensor(self.data).to(*a, **k), self.orig_shape)


class Results:
    """One image's detections (reference engine/results.py:71-120)."""

    def __init__(self, orig_img, path, names, boxes=None, speed=None):
        self.orig_img = orig_img
        self.orig_shape = tuple(orig_img.shape[:2])
        self.boxes = Boxes(boxes, self.orig_shape) if boxes is not None else None
        self.masks = self.probs = self.keypoints = self.obb = None
        self.speed = speed or {"preprocess": None, "inference": None, "postprocess": None}
        self.names, self.path, self.save_dir = names, path, None

    def __len__(self):
        return len(self.boxes) if self.boxes is not None else 0

    def cpu(self):
        r = Results(self.orig_img, self.path, self.names, speed=self.speed)
        r.boxes = self.boxes.cpu() if self.boxes is not None else None
        return r

    def numpy(self):
        r = Results(self.orig_img, self.path, self.names, speed=self.speed)
        r.boxes = self.boxes.numpy() if self.boxes is not None else None
        return r

    def verbose(self):
        """'3 class0s, 1 class2, ' -- the per-image log string of reference engine/results.py:345-360."""
        if not len(self):
            return "(no detections), "
        cls = self.boxes.cls
        cls = cls.cpu().numpy() if isinstance(cls, torch.Tensor) else cls
        out = ""
        for c in np.unique(cls):
            n = int((cls == c).sum())
            out += f"{n} {self.names[int(c)]}{'s' * (n > 1)}, "
        return out

    def tojson(self):
        import json
        d = self.boxes.numpy().data if self.boxes is not None else np.zeros((0, 6))
        return json.dumps([{"name": self.names[int(r[5])], "class": int(r[5]), "confidence": round(float(r[4]), 5),
                            "box": {"x1": float(r[0]), "y1": float(r[1]), "x2": float(r[2]), "y2": float(r[3])}} for r in d])
