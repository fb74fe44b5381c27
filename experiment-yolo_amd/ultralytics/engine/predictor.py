"""Prediction pipeline of the detect task (drop-in for reference engine/predictor.py:117-154, 243-330 and
models/yolo/detect/predict.py:23-43): source -> LetterBox -> fused-or-not forward on the HIP engine -> decode + soft-NMS kernels
-> boxes scaled back to the original image -> ``Results``.

Sources: an image file, a directory or glob of images, a list of those, PIL images, numpy HWC arrays (BGR, as cv2 would hand
them to the reference), or a (B,3,H,W) float tensor in [0,1].  Streams / videos / URLs, drawing and saving are control plane."""
from __future__ import annotations

import glob
import math
import os
import time
from pathlib import Path

import numpy as np
import torch

from ..cfg import get_cfg
from ..data.dataset import _resize_bilinear
from ..utils import LOGGER, ops
from .results import Results

IMG_FORMATS = {"bmp", "dng", "jpeg", "jpg", "mpo", "png", "tif", "tiff", "webp", "pfm"}  # reference data/utils.py:38


def load_source(source):
    """-> (paths, list of (H,W,3) uint8 RGB arrays) or (None, tensor) for a BCHW tensor (reference data/loaders.py + build.py:150-190)."""
    if isinstance(source, torch.Tensor):
        if source.dim() == 3:
            source = source[None]
        if source.dim() != 4 or source.shape[1] != 3:
            raise ValueError(f"tensor sources must be (B,3,H,W) in [0,1], got {tuple(source.shape)}")
        return [f"image{i}.jpg" for i in range(source.shape[0])], source
    items = list(source) if isinstance(source, (list, tuple)) else [source]
    paths, imgs = [], []
    for k, it in enumerate(items):
        if isinstance(it, (str, Path)):
            p = str(it)
            if os.path.isdir(p):
                files = sorted(glob.glob(os.path.join(p, "*.*")))
            elif "*" in p:
                files = sorted(glob.glob(p, recursive=True))
            elif os.path.isfile(p):
                files = [p]
            else:
                raise FileNotFoundError(f"{p} does not exist")
            files = [f for f in files if f.rsplit(".", 1)[-1].lower() in IMG_FORMATS]
            if not files:
                raise FileNotFoundError(f"No images found in {p}. Supported formats are: {sorted(IMG_FORMATS)}")
            from PIL import Image
            for f in files:
                paths.append(f)
                imgs.append(np.asarray(Image.open(f).convert("RGB")))
        elif isinstance(it, np.ndarray):  # HWC BGR like cv2.imread (reference LoadPilAndNumpy keeps numpy as is = BGR)
            if it.ndim != 3 or it.shape[2] != 3:
                raise ValueError(f"numpy sources must be (H,W,3) BGR uint8, got {it.shape}")
            paths.append(f"image{k}.jpg")
            imgs.append(np.ascontiguousarray(it[..., ::-1]))
        elif hasattr(it, "convert"):  # PIL
            paths.append(getattr(it, "filename", "") or f"image{k}.jpg")
            imgs.append(np.asarray(it.convert("RGB")))
        else:
            raise TypeError(f"unsupported prediction source {type(it).__name__} (file, directory, glob, PIL image, BGR ndarray or BCHW tensor)")
    return paths, imgs


def letterbox(img, new_shape, auto=False, stride=32, scaleup=True):
    """LetterBox.__call__ (reference data/augment.py:696-735, center=True, scaleFill=False) on an RGB uint8 array: resize so the
    image fits ``new_shape`` (bilinear; cv2's fixed-point INTER_LINEAR is not reproduced), pad with 114; ``auto`` = minimum
    rectangle (pads only up to the next multiple of ``stride``)."""
    shape = img.shape[:2]
    if isinstance(new_shape, int):
        new_shape = (new_shape, new_shape)
    r = min(new_shape[0] / shape[0], new_shape[1] / shape[1])
    if not scaleup:
        r = min(r, 1.0)
    new_unpad = int(round(shape[1] * r)), int(round(shape[0] * r))
    dw, dh = new_shape[1] - new_unpad[0], new_shape[0] - new_unpad[1]
    if auto:
        dw, dh = dw % stride, dh % stride
    dw, dh = dw / 2, dh / 2
    if shape[::-1] != new_unpad:
        img = _resize_bilinear(img, new_unpad[0], new_unpad[1])
    top, bottom = int(round(dh - 0.1)), int(round(dh + 0.1))
    left, right = int(round(dw - 0.1)), int(round(dw + 0.1))
    out = np.full((img.shape[0] + top + bottom, img.shape[1] + left + right, 3), 114, dtype=np.uint8)
    out[top:top + img.shape[0], left:left + img.shape[1]] = img
    return out


class DetectionPredictor:
    """reference models/yolo/detect/predict.py + engine/predictor.py, batch size = ``args.batch`` images per forward."""

    def __init__(self, cfg=None, overrides=None, _callbacks=None):
        self.args = get_cfg(overrides=overrides) if cfg is None else get_cfg(cfg, overrides)
        if self.args.conf is None:
            self.args.conf = 0.25  # predictor default (engine/predictor.py:88-89)
        self.model = self.device = None
        self.pt = False  # model came from a checkpoint: LetterBox(auto=True) for same-shaped batches (predictor.py:152)
        self.results, self.batch = None, None

    def setup_model(self, model, pt=False):
        from ..utils.torch_utils import select_device
        self.device = select_device(self.args.device)
        self.model = model.to(self.device).eval()
        self.pt = bool(pt)
        # reference nn/autobackend.py: stride = max(int(model.stride.max()), 32)
        self.stride = max(int(max(float(s) for s in model.stride)), 32) if hasattr(model, "stride") else 32
        s = self.args.imgsz
        s = [s, s] if isinstance(s, int) else list(s)
        self.imgsz = [max(math.ceil(x / self.stride) * self.stride, self.stride) for x in s]  # check_imgsz

    def pre_transform(self, imgs):
        same = all(x.shape == imgs[0].shape for x in imgs)
        return [letterbox(x, self.imgsz, auto=same and self.pt, stride=self.stride) for x in imgs]

    def preprocess(self, im):
        if not isinstance(im, torch.Tensor):
            im = torch.from_numpy(np.ascontiguousarray(np.stack(self.pre_transform(im)).transpose(0, 3, 1, 2)))  # RGB already
            return im.to(self.device).float() / 255
        return im.to(self.device).float()

    def postprocess(self, preds, img, orig_imgs, paths):
        a = self.args
        preds = ops.non_max_suppression(preds, a.conf, a.iou, agnostic=a.agnostic_nms, max_det=a.max_det, classes=a.classes)
        if isinstance(orig_imgs, torch.Tensor):  # ops.convert_torch2numpy_batch: BCHW [0,1] -> list of HWC uint8
            orig_imgs = list((orig_imgs.permute(0, 2, 3, 1).contiguous() * 255).to(torch.uint8).cpu().numpy())
        out = []
        for i, pred in enumerate(preds):
            orig = orig_imgs[i]
            pred[:, :4] = ops.scale_boxes(img.shape[2:], pred[:, :4], orig.shape)
            out.append(Results(orig, path=paths[i], names=self.model.names, boxes=pred))
        return out

    @torch.no_grad()
    def __call__(self, source=None, model=None, stream=False):
        if stream:
            raise NotImplementedError("stream=True generators are control plane; call predict per batch")
        if source is None:
            raise ValueError("predict needs a source (image file, directory, glob, list, PIL image, BGR ndarray or BCHW tensor)")
        paths, data = load_source(source)
        n = len(paths)
        bs = n if isinstance(data, torch.Tensor) else max(1, int(self.args.batch or 1))
        results = []
        for lo in range(0, n, bs):
            chunk = data[lo:lo + bs]
            t0 = time.perf_counter()
            im = self.preprocess(chunk)
            t1 = time.perf_counter()
            preds = self.model(im)
            if self.args.verbose:
                torch.cuda.synchronize()
            t2 = time.perf_counter()
            res = self.postprocess(preds, im, chunk, paths[lo:lo + bs])
            t3 = time.perf_counter()
            for k, r in enumerate(res):
                r.speed = {"preprocess": (t1 - t0) * 1e3 / len(res), "inference": (t2 - t1) * 1e3 / len(res), "postprocess": (t3 - t2) * 1e3 / len(res)}
                if self.args.verbose:
                    LOGGER.info(f"image {lo + k + 1}/{n} {r.path}: {im.shape[2]}x{im.shape[3]} {r.verbose()}{r.speed['inference']:.1f}ms")
            results += res
        if self.args.save_txt or self.args.save_crop or self.args.save_conf:
            LOGGER.warning("WARNING save_txt / save_crop / save_conf: writing results is control plane, not provided by this package")
        self.results = results
        return results

    predict_cli = __call__
