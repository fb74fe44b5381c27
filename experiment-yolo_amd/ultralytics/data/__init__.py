"""Batch sources in the dataloader's batch-dict format (reference data/dataset.py:207-224): the YOLO-format dataset reader
and loader (dataset.py / build.py / utils.py, SURVEY.md section 8f row 2 -- augmentation-free subset) and a synthetic source
following the recipe of SURVEY.md section 8(d)."""
import numpy as np
import torch

from .build import HipDataLoader, build_dataloader, build_yolo_dataset  # noqa: F401
from .dataset import YOLODataset  # noqa: F401
from .utils import check_det_dataset  # noqa: F401


class SyntheticDetection:
    def __init__(self, n_batches, batch, imgsz=640, nc=6, boxes_per_image=8, seed=1, wh=(0.01, 0.09), device="cpu"):
        self.n, self.B, self.s, self.nc, self.k, self.seed, self.wh, self.device = n_batches, batch, imgsz, nc, boxes_per_image, seed, wh, device

    def __len__(self):
        return self.n

    def __iter__(self):
        dev = torch.device(self.device)
        for i in range(self.n):
            rng = np.random.default_rng(self.seed * 100003 + i)
            n = self.B * self.k
            if dev.type == "cuda":  # images drawn on the device (a 64 x 3 x 640 x 640 fp32 batch takes ~100 ms of host RNG otherwise)
                g = torch.Generator(device=dev).manual_seed(self.seed * 100003 + i)
                img = torch.rand((self.B, 3, self.s, self.s), generator=g, device=dev)
            else:
                img = torch.from_numpy(rng.random((self.B, 3, self.s, self.s), dtype=np.float32))
            yield {k: v.to(dev) for k, v in dict(
                img=img,
                batch_idx=torch.arange(self.B).repeat_interleave(self.k).float(),
                cls=torch.from_numpy(rng.integers(0, self.nc, (n, 1)).astype(np.float32)),
                bboxes=torch.from_numpy(np.concatenate([rng.random((n, 2)) * 0.8 + 0.1, rng.random((n, 2)) * (self.wh[1] - self.wh[0]) + self.wh[0]], 1).astype(np.float32))).items()}
