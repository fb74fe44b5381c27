"""YOLO-format detection dataset (drop-in for the detect-task, augmentation-free subset of reference data/base.py:21-330 +
data/dataset.py:23-224 + the LetterBox / RandomPerspective(zero gains) / Format transforms of data/augment.py).

A sample is the reference's dict -- ``im_file, ori_shape, resized_shape, ratio_pad (val), img, cls (n,1), bboxes (n,4)
normalised xywh, batch_idx`` -- with one difference chosen for the device path: ``img`` is uint8 **HWC RGB** (what the
import kernel ``dy_import_image_u8`` turns into the fp16 NHWC input of the stem), unless ``layout="nchw"`` asks for the
reference's CHW tensor.  Geometry and label arithmetic follow the reference operation by operation in float32:

* ``load_image`` (base.py:146-181): ``*.npy`` sibling (BGR) preferred, else the image file; long side resized to ``imgsz``
  with bilinear interpolation (the reference: cv2 INTER_LINEAR, fixed-point -- the only step that is not bit-pinned);
* train mode = the reference's augment pipeline with every gain at zero (mosaic/mixup/copy_paste/hsv/degrees/translate/
  scale/shear/perspective/flip = 0): LetterBox(scaleup=True) to (imgsz, imgsz), boxes clipped to the canvas, then
  RandomPerspective.box_candidates (:562-581) drops boxes under 2 px, with aspect ratio >= 100 or that lost 90 % of their area;
* val mode: rectangular batches (base.py:224-246: aspect-sorted, per-batch shape ceil(shape*imgsz/stride + 0.5)*stride) and
  LetterBox(scaleup=False) to the batch shape; ``ratio_pad`` kept for the validator's box rescaling.
Of the augmentations, the two flips are on this path (``flipud`` / ``fliplr`` probabilities): the decisions come from
Python's ``random`` in the reference's per-sample draw order (Mosaic's probability draw, the eight RandomPerspective draws and
MixUp's draw are consumed even at zero gain, augment.py:105, :406-425), so a run seeded like the reference flips the same
images; labels flip on the host, pixels either on the host or -- ``flip_on_device`` -- inside the import kernel.  Mosaic, HSV
and the affine warp (cv2 + numpy RNG) are not on this path."""
from __future__ import annotations

import glob
import math
import os
import random
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

import numpy as np
import torch
from PIL import Image

from ..utils import LOGGER
from .utils import IMG_FORMATS, img2label_paths, verify_image_label


def _resize_bilinear(im: np.ndarray, w: int, h: int) -> np.ndarray:
    t = torch.from_numpy(np.array(im)).permute(2, 0, 1)[None].float()
    t = torch.nn.functional.interpolate(t, size=(h, w), mode="bilinear", align_corners=False)
    return t.round_().clamp_(0, 255).byte()[0].permute(1, 2, 0).contiguous().numpy()


def letterbox_geometry(shape, new_shape, scaleup):
    """LetterBox.__call__ (augment.py:696-735, center=True): -> r, (dw, dh) float halves, (top, bottom, left, right)."""
    r = min(new_shape[0] / shape[0], new_shape[1] / shape[1])
    if not scaleup:
        r = min(r, 1.0)
    new_unpad = int(round(shape[1] * r)), int(round(shape[0] * r))
    dw, dh = (new_shape[1] - new_unpad[0]) / 2, (new_shape[0] - new_unpad[1]) / 2
    return r, new_unpad, (dw, dh), (int(round(dh - 0.1)), int(round(dh + 0.1)), int(round(dw - 0.1)), int(round(dw + 0.1)))


class YOLODataset:
    def __init__(self, img_path, imgsz=640, batch_size=16, augment=False, rect=False, stride=32, pad=0.0, data=None, fraction=1.0,
                 cache=False, layout="nhwc", prefix="", flipud=0.0, fliplr=0.0, flip_on_device=False):
        self.img_path, self.imgsz, self.batch_size, self.augment, self.rect = img_path, int(imgsz), batch_size, augment, rect
        self.flipud, self.fliplr, self.flip_on_device = float(flipud), float(fliplr), flip_on_device
        self.stride, self.pad, self.data, self.fraction, self.layout, self.prefix = stride, pad, data or {}, fraction, layout, prefix
        self.im_files = self.get_img_files(img_path)
        self.labels = self.get_labels()
        self.ni = len(self.labels)
        if rect:
            self.set_rectangle()
        self.npy_files = [Path(f).with_suffix(".npy") for f in self.im_files]
        self.ims = [None] * self.ni
        self.im_hw0, self.im_hw = [None] * self.ni, [None] * self.ni  # shapes seen by load_image (label-only sample building)
        self.cache_mode = cache  # False | True/'ram' (decoded images in host RAM) | 'hbm' (the loader keeps them on the device)
        if cache and cache != "hbm":
            with ThreadPoolExecutor(min(8, os.cpu_count() or 1)) as pool:
                self.ims = list(pool.map(self.load_image, range(self.ni)))

    # ---- files & labels ------------------------------------------------------------------------------------------
    def get_img_files(self, img_path):
        f = []
        for p in img_path if isinstance(img_path, list) else [img_path]:
            p = Path(p)
            if p.is_dir():
                f += glob.glob(str(p / "**" / "*.*"), recursive=True)
            elif p.is_file():
                with open(p) as t:
                    parent = str(p.parent) + os.sep
                    f += [x.replace("./", parent) if x.startswith("./") else x for x in t.read().strip().splitlines()]
            else:
                raise FileNotFoundError(f"{self.prefix}{p} does not exist")
        im_files = sorted(x for x in f if x.split(".")[-1].lower() in IMG_FORMATS)
        if not im_files:
            raise FileNotFoundError(f"{self.prefix}No images found in {img_path}")
        if self.fraction < 1:
            im_files = im_files[: round(len(im_files) * self.fraction)]
        return im_files

    def get_labels(self):
        """dataset.py:44-150 without the on-disk *.cache file: every pair is verified on each construction (threads)."""
        label_files = img2label_paths(self.im_files)
        nc = len(self.data["names"]) if "names" in self.data else int(self.data.get("nc", 1 << 30))
        with ThreadPoolExecutor(min(8, os.cpu_count() or 1)) as pool:
            res = list(pool.map(lambda a: verify_image_label(a[0], a[1], nc), zip(self.im_files, label_files)))
        labels, nm, nf, ne, ncor = [], 0, 0, 0, 0
        for im_file, lb, shape, m, f, e, c, msg in res:
            nm, nf, ne, ncor = nm + m, nf + f, ne + e, ncor + c
            if im_file:
                labels.append(dict(im_file=im_file, shape=shape, cls=lb[:, 0:1], bboxes=lb[:, 1:]))
            if msg:
                LOGGER.info(self.prefix + msg)
        LOGGER.info(f"{self.prefix}{nf} images, {nm + ne} backgrounds, {ncor} corrupt")
        if not labels:
            LOGGER.warning(f"{self.prefix}WARNING No images found, training may not work correctly.")
        self.im_files = [lb["im_file"] for lb in labels]
        return labels

    def set_rectangle(self):
        """base.py:224-246."""
        bi = np.floor(np.arange(self.ni) / self.batch_size).astype(int)
        nb = bi[-1] + 1
        s = np.array([x["shape"] for x in self.labels])  # hw
        ar = s[:, 0] / s[:, 1]
        irect = ar.argsort()
        self.im_files = [self.im_files[i] for i in irect]
        self.labels = [self.labels[i] for i in irect]
        ar = ar[irect]
        shapes = [[1, 1]] * nb
        for i in range(nb):
            ari = ar[bi == i]
            mini, maxi = ari.min(), ari.max()
            if maxi < 1:
                shapes[i] = [maxi, 1]
            elif mini > 1:
                shapes[i] = [1, 1 / mini]
        self.batch_shapes = np.ceil(np.array(shapes) * self.imgsz / self.stride + self.pad).astype(int) * self.stride
        self.batch = bi

    # ---- samples -------------------------------------------------------------------------------------------------
    def load_image(self, i):
        """-> (uint8 HWC **RGB**, (h0, w0), (h, w)); long side == imgsz afterwards (base.py:146-181, rect_mode=True)."""
        if self.ims[i] is not None:
            return self.ims[i]
        f, fn = self.im_files[i], self.npy_files[i]
        if fn.exists():
            im = np.ascontiguousarray(np.load(fn)[..., ::-1])  # the reference's *.npy caches hold BGR
        else:
            im = np.asarray(Image.open(f).convert("RGB"))
        h0, w0 = im.shape[:2]
        r = self.imgsz / max(h0, w0)
        if r != 1:
            w, h = min(math.ceil(w0 * r), self.imgsz), min(math.ceil(h0 * r), self.imgsz)
            im = _resize_bilinear(im, w, h)
        return im, (h0, w0), im.shape[:2]

    def __len__(self):
        return self.ni

    def draw_augment(self):
        """The random decisions of one training sample, drawn from Python's ``random`` in the reference's order (call this in
        sample order from ONE thread; the pixels can then be produced by any worker).  -> flip bits: 1 = left-right, 2 = up-down."""
        if not self.augment:
            return 0
        random.uniform(0, 1)                                  # Mosaic.__call__ probability check   augment.py:105
        for _ in range(8):                                    # RandomPerspective.affine_transform  :406-425
            random.uniform(0, 0)
        random.uniform(0, 1)                                  # MixUp.__call__ probability check    :105
        ud = random.random() < self.flipud                    # RandomFlip(vertical)                :670
        lr = random.random() < self.fliplr                    # RandomFlip(horizontal)              :674
        return (1 if lr else 0) | (2 if ud else 0)

    def __getitem__(self, index):
        return self.get(index, self.draw_augment())

    def get(self, index, flip=0, pixels=True):
        """One sample; ``pixels=False`` builds the labels only (the image already sits in the loader's HBM pool)."""
        lab = self.labels[index]
        if pixels or self.im_hw[index] is None:
            im, ori_shape, resized = self.load_image(index)
            self.im_hw0[index], self.im_hw[index] = ori_shape, resized
        else:
            im, ori_shape, resized = None, self.im_hw0[index], self.im_hw[index]
        h, w = resized
        new_shape = tuple(int(v) for v in self.batch_shapes[self.batch[index]]) if self.rect else (self.imgsz, self.imgsz)
        r, new_unpad, (dw, dh), (top, bottom, left, right) = letterbox_geometry((h, w), new_shape, scaleup=self.augment)
        if im is not None and (w, h) != new_unpad:
            im = _resize_bilinear(im, *new_unpad)
        H, W = new_unpad[1] + top + bottom, new_unpad[0] + left + right
        if im is None:
            canvas = None
        elif top or bottom or left or right:
            canvas = np.full((H, W, 3), 114, dtype=np.uint8)
            canvas[top:top + im.shape[0], left:left + im.shape[1]] = im
        else:
            canvas = im  # already the canvas size: handed on without a copy (the loader copies it into its pinned batch)
        # labels: normalised xywh -> xyxy -> pixels of the (resized) image -> * r -> + pad   (LetterBox._update_labels :744-750)
        b = lab["bboxes"].astype(np.float32, copy=True)
        cls = lab["cls"].copy()
        xy = np.empty_like(b)
        hw_, hh_ = b[:, 2] / 2, b[:, 3] / 2
        xy[:, 0], xy[:, 1], xy[:, 2], xy[:, 3] = b[:, 0] - hw_, b[:, 1] - hh_, b[:, 0] + hw_, b[:, 1] + hh_
        for j, sc in enumerate((w, h, w, h)):
            xy[:, j] *= sc
        for j in range(4):
            xy[:, j] *= r
        for j, off in enumerate((dw, dh, dw, dh)):
            xy[:, j] += off
        if self.augment:  # RandomPerspective with the identity matrix (:512-560): clip, then filter the candidates
            before = xy.copy()
            xy[:, [0, 2]] = xy[:, [0, 2]].clip(0, W)
            xy[:, [1, 3]] = xy[:, [1, 3]].clip(0, H)
            w1, h1 = before[:, 2] - before[:, 0], before[:, 3] - before[:, 1]
            w2, h2 = xy[:, 2] - xy[:, 0], xy[:, 3] - xy[:, 1]
            eps = np.float32(1e-16)
            ar = np.maximum(w2 / (h2 + eps), h2 / (w2 + eps))
            keep = (w2 > 2) & (h2 > 2) & (w2 * h2 / (w1 * h1 + eps) > np.float32(0.1)) & (ar < 100)
            xy, cls = xy[keep], cls[keep]
        out = np.empty_like(xy)  # RandomFlip / Format (:664, :918): xyxy -> xywh; flips mirror the centre; normalise by the canvas
        out[:, 0], out[:, 1] = (xy[:, 0] + xy[:, 2]) / 2, (xy[:, 1] + xy[:, 3]) / 2
        out[:, 2], out[:, 3] = xy[:, 2] - xy[:, 0], xy[:, 3] - xy[:, 1]
        if flip & 2:
            out[:, 1] = H - out[:, 1]
        if flip & 1:
            out[:, 0] = W - out[:, 0]
        if flip and not self.flip_on_device and canvas is not None:
            canvas = np.ascontiguousarray(canvas[::-1 if flip & 2 else 1, ::-1 if flip & 1 else 1])
        for j, sc in enumerate((1 / W, 1 / H, 1 / W, 1 / H)):
            out[:, j] *= sc
        nl = len(out)
        s = dict(im_file=lab["im_file"], ori_shape=ori_shape, resized_shape=(H, W) if self.augment else new_shape)
        if not self.augment:
            s["ratio_pad"] = ((h / ori_shape[0], w / ori_shape[1]), (left, top))
        if canvas is not None:
            s["img"] = torch.from_numpy(canvas if self.layout == "nhwc" else np.ascontiguousarray(canvas.transpose(2, 0, 1)))
        s["cls"] = torch.from_numpy(cls) if nl else torch.zeros(nl)
        s["bboxes"] = torch.from_numpy(out) if nl else torch.zeros((nl, 4))
        s["batch_idx"] = torch.zeros(nl)
        if self.flip_on_device:
            s["flip"] = flip  # pixels are delivered unflipped: dy_import_image_u8 mirrors them while converting
        return s

    @staticmethod
    def collate_fn(batch):
        """dataset.py:207-224."""
        new = {}
        for k in batch[0].keys():
            value = [b[k] for b in batch]
            if k == "img":
                value = torch.stack(value, 0)
            if k == "flip":
                value = torch.tensor(value, dtype=torch.uint8)
            if k in ("bboxes", "cls"):
                value = torch.cat([v.reshape(-1, 4 if k == "bboxes" else 1) if v.numel() == 0 else v for v in value], 0)
            new[k] = value
        bi = list(new["batch_idx"])
        for i in range(len(bi)):
            bi[i] = bi[i] + i
        new["batch_idx"] = torch.cat(bi, 0)
        return new
