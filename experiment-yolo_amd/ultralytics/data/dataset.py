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
                 cache=False, layout="nhwc", prefix="", flipud=0.0, fliplr=0.0, flip_on_device=False, mosaic=0.0, degrees=0.0,
                 translate=0.0, scale=0.0, shear=0.0, hsv_h=0.0, hsv_s=0.0, hsv_v=0.0, perspective=0.0, mixup=0.0, copy_paste=0.0):
        self.img_path, self.imgsz, self.batch_size, self.augment, self.rect = img_path, int(imgsz), batch_size, augment, rect
        self.flipud, self.fliplr, self.flip_on_device = float(flipud), float(fliplr), flip_on_device
        self.mosaic, self.degrees, self.translate, self.scale, self.shear = (float(v) for v in (mosaic, degrees, translate, scale, shear))
        self.perspective, self.mixup = float(perspective), float(mixup)
        # copy_paste: the reference's CopyPaste only acts on segment labels (augment.py:`if self.p and len(instances.segments)`) and
        # draws nothing before that test, so for box-only detection labels it is a no-op at any probability -- accepted and ignored
        self.copy_paste = float(copy_paste)
        self.geometric = augment and any((self.mosaic, self.degrees, self.translate, self.scale, self.shear, self.perspective, self.mixup))
        self.hsv = (float(hsv_h), float(hsv_s), float(hsv_v)) if augment and any((hsv_h, hsv_s, hsv_v)) else None
        self._label_cache = {}
        self.buffer, self._in_buffer = [], set()  # BaseDataset.buffer (base.py:86-87, :170-176): what Mosaic draws its partners from
        self.stride, self.pad, self.data, self.fraction, self.layout, self.prefix = stride, pad, data or {}, fraction, layout, prefix
        self.im_files = self.get_img_files(img_path)
        self.labels = self.get_labels()
        self.ni = len(self.labels)
        self.max_buffer_length = min(self.ni, self.batch_size * 8, 1000) if augment else 0
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

    def _touch(self, i):
        """The bookkeeping side of the reference's load_image under augmentation (base.py:170-176): an image that is not in the
        RAM buffer enters it, and the oldest one leaves when the buffer is full."""
        if i in self._in_buffer:
            return
        self._in_buffer.add(i)
        self.buffer.append(i)
        if len(self.buffer) >= self.max_buffer_length:
            self._in_buffer.discard(self.buffer.pop(0))

    def _draw_pre(self, index):
        """Draws of the reference's ``pre_transform`` = Compose([Mosaic, CopyPaste, RandomPerspective]) for ONE sample, in its order
        (augment.py:105-110, 159-162, 212, 406-425); ``get_image_and_label(index)`` ran before (buffer bookkeeping)."""
        self._touch(index)
        aug = {"mosaic": None}
        if random.uniform(0, 1) <= self.mosaic:                      # Mosaic.__call__ (augment.py:105): skipped when u > p
            partners = random.choices(list(self.buffer), k=3)        # get_indexes(buffer=True) :159-162
            for i in partners:
                self._touch(i)
            b = -self.imgsz // 2
            yc = int(random.uniform(-b, 2 * self.imgsz + b))          # _mosaic4 :212 (y first)
            xc = int(random.uniform(-b, 2 * self.imgsz + b))
            aug["mosaic"] = (partners, yc, xc)
        px = random.uniform(-self.perspective, self.perspective)    # RandomPerspective.affine_transform :406-407
        py = random.uniform(-self.perspective, self.perspective)
        a = random.uniform(-self.degrees, self.degrees)       # :411
        sc = random.uniform(1 - self.scale, 1 + self.scale)   # :413
        shx = math.tan(random.uniform(-self.shear, self.shear) * math.pi / 180)   # :419-420
        shy = math.tan(random.uniform(-self.shear, self.shear) * math.pi / 180)
        tx = random.uniform(0.5 - self.translate, 0.5 + self.translate)           # :424-425
        ty = random.uniform(0.5 - self.translate, 0.5 + self.translate)
        aug["affine"], aug["persp"] = (a, sc, shx, shy, tx, ty), (px, py)
        return aug

    def draw_augment(self, index=None):
        """The random decisions of one training sample, drawn from Python's ``random`` (and numpy's global RNG for MixUp's ratio
        and the HSV gains) in the reference's order (call this in sample order from ONE thread; pixels and labels can then be
        produced by any worker).  Returns the flip bits (1 = left-right, 2 = up-down) or, when a geometric augmentation is on, a
        dict with them plus the mosaic partners / centre, the affine + perspective parameters and the MixUp partner."""
        if not self.augment:
            return 0
        aug = None
        if self.geometric:
            aug = self._draw_pre(index)
            if random.uniform(0, 1) <= self.mixup:             # MixUp.__call__ (BaseMixTransform :105): mixes unless u > p
                i2 = random.randint(0, len(self) - 1)          # MixUp.get_indexes :336
                aug2 = self._draw_pre(i2)                      # get_image_and_label(i2), then pre_transform on it :112-116
                aug["mix"] = (i2, aug2, float(np.random.beta(32.0, 32.0)))   # _mix_transform :340
        else:
            random.uniform(0, 1)                               # Mosaic's probability draw (p = 0)
            for _ in range(8):                                 # RandomPerspective's eight draws (all ranges empty)
                random.uniform(0, 0)
            random.uniform(0, 1)                               # MixUp's probability draw (p = 0)
        gains = None
        if self.hsv is not None:                              # RandomHSV :613 -- numpy's global RNG, three draws per sample
            gains = (np.random.uniform(-1, 1, 3) * self.hsv + 1).astype(np.float32)
        ud = random.random() < self.flipud                    # RandomFlip(vertical)                :670
        lr = random.random() < self.fliplr                    # RandomFlip(horizontal)              :674
        flip = (1 if lr else 0) | (2 if ud else 0)
        if aug is None and gains is None:
            return flip
        aug = aug if aug is not None else {}
        aug.update(flip=flip, hsv=gains)
        return aug

    def __getitem__(self, index):
        return self.get(index, self.draw_augment(index))

    def _hw(self, i):
        if self.im_hw[i] is None:
            _, self.im_hw0[i], self.im_hw[i] = self.load_image(i)
        return self.im_hw[i]

    @staticmethod
    def _pixel_boxes(lab, w, h, padw, padh):
        """normalised xywh -> xyxy -> * (w, h) -> + pad, column by column in float32 (Mosaic._update_labels :292-299)."""
        b = lab["bboxes"].astype(np.float32, copy=True)
        xy = np.empty_like(b)
        hw_, hh_ = b[:, 2] / 2, b[:, 3] / 2
        xy[:, 0], xy[:, 1], xy[:, 2], xy[:, 3] = b[:, 0] - hw_, b[:, 1] - hh_, b[:, 0] + hw_, b[:, 1] + hh_
        for j, sc in enumerate((w, h, w, h)):
            xy[:, j] *= sc
        for j, off in enumerate((padw, padh, padw, padh)):
            xy[:, j] += off
        return xy

    def mosaic_layout(self, index, aug):
        """The four placements of Mosaic._mosaic4 (:208-241) on the (2s x 2s) canvas: per patch (image index, destination
        x1a, y1a, x2a, y2a, source x1b, y1b)."""
        partners, yc, xc = aug["mosaic"]
        s2 = self.imgsz * 2
        out = []
        for k, i in enumerate([index, *partners]):
            h, w = self._hw(i)
            if k == 0:
                x1a, y1a, x2a, y2a = max(xc - w, 0), max(yc - h, 0), xc, yc
                x1b, y1b = w - (x2a - x1a), h - (y2a - y1a)
            elif k == 1:
                x1a, y1a, x2a, y2a = xc, max(yc - h, 0), min(xc + w, s2), yc
                x1b, y1b = 0, h - (y2a - y1a)
            elif k == 2:
                x1a, y1a, x2a, y2a = max(xc - w, 0), yc, xc, min(s2, yc + h)
                x1b, y1b = w - (x2a - x1a), 0
            else:
                x1a, y1a, x2a, y2a = xc, yc, min(xc + w, s2), min(s2, yc + h)
                x1b, y1b = 0, 0
            out.append((i, x1a, y1a, x2a, y2a, x1b, y1b))
        return out

    def affine_matrix(self, aug, img_w, img_h, size):
        """RandomPerspective.affine_transform (:384-435): M = T @ S @ R @ P @ C in float32."""
        a, sc, shx, shy, tx, ty = aug["affine"]
        Cm = np.eye(3, dtype=np.float32)
        Cm[0, 2], Cm[1, 2] = -img_w / 2, -img_h / 2
        P = np.eye(3, dtype=np.float32)
        P[2, 0], P[2, 1] = aug.get("persp", (0.0, 0.0))
        R = np.eye(3, dtype=np.float32)
        al, be = sc * math.cos(math.radians(a)), sc * math.sin(math.radians(a))  # cv2.getRotationMatrix2D(angle, (0, 0), scale)
        R[:2] = np.array([[al, be, 0.0], [-be, al, 0.0]])
        S = np.eye(3, dtype=np.float32)
        S[0, 1], S[1, 0] = shx, shy
        T = np.eye(3, dtype=np.float32)
        T[0, 2], T[1, 2] = tx * size[0], ty * size[1]
        return T @ S @ R @ P @ Cm

    def _geo_labels(self, index, aug):
        """Labels through Mosaic / LetterBox and RandomPerspective (:512-560) -> (xyxy in canvas pixels, cls, (W, H), M)."""
        s = self.imgsz
        if aug["mosaic"] is not None:
            parts, clss = [], []
            for i, x1a, y1a, x2a, y2a, x1b, y1b in self.mosaic_layout(index, aug):
                h, w = self._hw(i)
                parts.append(self._pixel_boxes(self.labels[i], w, h, x1a - x1b, y1a - y1b))
                clss.append(self.labels[i]["cls"])
            xy, cls = np.concatenate(parts, 0), np.concatenate(clss, 0)
            xy[:, [0, 2]] = xy[:, [0, 2]].clip(0, 2 * s)       # _cat_labels :318-319: clip to the mosaic, drop empty boxes
            xy[:, [1, 3]] = xy[:, [1, 3]].clip(0, 2 * s)
            good = (xy[:, 2] - xy[:, 0]) * (xy[:, 3] - xy[:, 1]) > 0
            xy, cls = xy[good], cls[good]
            img_w = img_h = 2 * s
            border = (-s // 2, -s // 2)
        else:
            h, w = self._hw(index)
            r, new_unpad, (dw, dh), (top, bottom, left, right) = letterbox_geometry((h, w), (s, s), scaleup=True)
            xy = self._pixel_boxes(self.labels[index], w, h, 0, 0)
            for j in range(4):
                xy[:, j] *= r
            for j, off in enumerate((dw, dh, dw, dh)):
                xy[:, j] += off
            cls = self.labels[index]["cls"].copy()
            img_w, img_h = new_unpad[0] + left + right, new_unpad[1] + top + bottom
            border = (0, 0)
        size = (img_w + border[1] * 2, img_h + border[0] * 2)
        M = self.affine_matrix(aug, img_w, img_h, size)
        n = len(xy)
        if n:
            pts = np.ones((n * 4, 3), dtype=xy.dtype)
            pts[:, :2] = xy[:, [0, 1, 2, 3, 0, 3, 2, 1]].reshape(n * 4, 2)  # x1y1, x2y2, x1y2, x2y1
            pts = pts @ M.T
            pts = (pts[:, :2] / pts[:, 2:3] if self.perspective else pts[:, :2]).reshape(n, 8)  # apply_bboxes :454
            px, py = pts[:, [0, 2, 4, 6]], pts[:, [1, 3, 5, 7]]
            new = np.concatenate((px.min(1), py.min(1), px.max(1), py.max(1)), dtype=xy.dtype).reshape(4, n).T
        else:
            new = xy
        new[:, [0, 2]] = new[:, [0, 2]].clip(0, size[0])
        new[:, [1, 3]] = new[:, [1, 3]].clip(0, size[1])
        sc = aug["affine"][1]
        for j in range(4):
            xy[:, j] *= sc                                       # instances.scale(scale, scale, bbox_only=True) :551
        w1, h1 = xy[:, 2] - xy[:, 0], xy[:, 3] - xy[:, 1]
        w2, h2 = new[:, 2] - new[:, 0], new[:, 3] - new[:, 1]
        eps = np.float32(1e-16)
        ar = np.maximum(w2 / (h2 + eps), h2 / (w2 + eps))
        keep = (w2 > 2) & (h2 > 2) & (w2 * h2 / (w1 * h1 + eps) > np.float32(0.1)) & (ar < 100)
        return new[keep], cls[keep], size, M

    def _letterbox_labels(self, lab, w, h, r, dw, dh, W, H):
        """normalised xywh -> xyxy -> pixels of the (resized) image -> * r -> + pad (LetterBox._update_labels :744-750); in train
        mode the identity RandomPerspective (:512-560: clip, candidate filter); -> pixel xywh (RandomFlip / Format :664, :918)."""
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
        if self.augment:
            before = xy.copy()
            xy[:, [0, 2]] = xy[:, [0, 2]].clip(0, W)
            xy[:, [1, 3]] = xy[:, [1, 3]].clip(0, H)
            w1, h1 = before[:, 2] - before[:, 0], before[:, 3] - before[:, 1]
            w2, h2 = xy[:, 2] - xy[:, 0], xy[:, 3] - xy[:, 1]
            eps = np.float32(1e-16)
            ar = np.maximum(w2 / (h2 + eps), h2 / (w2 + eps))
            keep = (w2 > 2) & (h2 > 2) & (w2 * h2 / (w1 * h1 + eps) > np.float32(0.1)) & (ar < 100)
            xy, cls = xy[keep], cls[keep]
        out = np.empty_like(xy)
        out[:, 0], out[:, 1] = (xy[:, 0] + xy[:, 2]) / 2, (xy[:, 1] + xy[:, 3]) / 2
        out[:, 2], out[:, 3] = xy[:, 2] - xy[:, 0], xy[:, 3] - xy[:, 1]
        return out, cls

    def get(self, index, flip=0, pixels=True):
        """One sample; ``pixels=False`` builds the labels only (the image already sits in the loader's HBM pool).  ``flip``: the
        value ``draw_augment`` returned (flip bits, or the dict of a geometric augmentation)."""
        gains = None
        if isinstance(flip, dict):
            if "affine" in flip:
                return self._get_geometric(index, flip, pixels)
            flip, gains = flip["flip"], flip["hsv"]  # colour jitter (+ flips) only
            if gains is not None and not self.flip_on_device:
                raise NotImplementedError("HSV jitter is applied inside the device import kernels (flip_on_device loaders)")
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
        # labels before the flips depend on the image alone: computed once per image, copied per sample
        cached = self._label_cache.get(index) if not self.rect else None
        if cached is None:
            cached = self._letterbox_labels(lab, w, h, r, dw, dh, W, H)
            if not self.rect:
                self._label_cache[index] = cached
        out, cls = cached[0].copy(), cached[1].copy()
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
            if self.hsv is not None:
                s["hsv"] = torch.from_numpy(gains if gains is not None else np.ones(3, np.float32))
        return s

    WARP_WORDS = 48  # one record; a sample carries two (itself + its MixUp partner, unused when it has none)

    def warp_slot(self, index, aug, M, mix_r=None):
        """The 48-word record dy_warp_import_u8 reads (see include/dealyolo_hip.h): inverse map (rows 0-1 in words 0-5, the
        perspective row in 43-45), canvas, mosaic centre, flip bits, up to four (pool image, destination rectangle, source corner)
        patches, HSV gains (40-42), MixUp ratio as float64 (46-47; negative: no partner).  Pool images are the letterboxed s x s
        canvases the loader uploaded, so a patch's source corner is shifted by that image's letterbox pad."""
        s = self.imgsz
        rec = np.zeros(self.WARP_WORDS, dtype=np.int32)
        if aug.get("hsv") is not None:
            rec[40:43] = np.asarray(aug["hsv"], np.float32).view(np.int32)
        minv = np.linalg.inv(M.astype(np.float64)).astype(np.float32)  # cv2.warpAffine / warpPerspective invert the forward map
        rec[:6] = minv[:2].reshape(-1).view(np.int32)
        rec[43:46] = (minv[2] if self.perspective else np.array([0, 0, 1], np.float32)).view(np.int32)
        rec[46:48] = np.array([-1.0 if mix_r is None else mix_r], np.float64).view(np.int32)

        def pad_of(i):
            h, w = self._hw(i)
            _, _, _, (top, _, left, _) = letterbox_geometry((h, w), (s, s), scaleup=True)
            return left, top

        if aug["mosaic"] is not None:
            _, yc, xc = aug["mosaic"]
            rec[6:12] = (2 * s, 2 * s, xc, yc, aug.get("flip", 0), 4)
            for k, (i, x1a, y1a, x2a, y2a, x1b, y1b) in enumerate(self.mosaic_layout(index, aug)):
                left, top = pad_of(i)
                rec[12 + 7 * k:19 + 7 * k] = (i, x1a, y1a, x2a, y2a, x1b + left, y1b + top)
        else:
            rec[6:12] = (s, s, s, s, aug.get("flip", 0), 1)
            rec[12:19] = (index, 0, 0, s, s, 0, 0)
        return rec

    def _get_geometric(self, index, aug, pixels):
        """Mosaic / affine / perspective / MixUp sample: labels on the host; the pixels of this path are composed on the device
        from the loader's HBM pool (the reference warps with cv2), so only ``pixels=False`` is served here."""
        if pixels:
            raise NotImplementedError("mosaic / affine pixels are composed on the device from the HBM image pool (cache='hbm')")
        lab = self.labels[index]
        xy, cls, (W, H), M = self._geo_labels(index, aug)
        mix = aug.get("mix")
        rec2 = np.zeros(self.WARP_WORDS, dtype=np.int32)
        if mix is not None:  # MixUp._mix_transform (augment.py:338-345): the partner went through the same pre_transform
            i2, aug2, r = mix
            xy2, cls2, _, M2 = self._geo_labels(i2, aug2)
            xy, cls = np.concatenate((xy, xy2), 0), np.concatenate((cls, cls2), 0)
            rec2 = self.warp_slot(i2, aug2, M2)
        flip = aug["flip"]
        out = np.empty_like(xy)
        out[:, 0], out[:, 1] = (xy[:, 0] + xy[:, 2]) / 2, (xy[:, 1] + xy[:, 3]) / 2
        out[:, 2], out[:, 3] = xy[:, 2] - xy[:, 0], xy[:, 3] - xy[:, 1]
        if flip & 2:
            out[:, 1] = H - out[:, 1]
        if flip & 1:
            out[:, 0] = W - out[:, 0]
        for j, sc in enumerate((1 / W, 1 / H, 1 / W, 1 / H)):
            out[:, j] *= sc
        nl = len(out)
        s = dict(im_file=lab["im_file"], ori_shape=self.im_hw0[index], resized_shape=(H, W))
        s["cls"] = torch.from_numpy(cls) if nl else torch.zeros(nl)
        s["bboxes"] = torch.from_numpy(out) if nl else torch.zeros((nl, 4))
        s["batch_idx"] = torch.zeros(nl)
        s["warp"] = torch.from_numpy(np.concatenate((self.warp_slot(index, aug, M, None if mix is None else mix[2]), rec2)))
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
            if k in ("warp", "hsv"):
                value = torch.stack(value, 0)
            if k in ("bboxes", "cls"):
                value = torch.cat([v.reshape(-1, 4 if k == "bboxes" else 1) if v.numel() == 0 else v for v in value], 0)
            new[k] = value
        bi = list(new["batch_idx"])
        for i in range(len(bi)):
            bi[i] = bi[i] + i
        new["batch_idx"] = torch.cat(bi, 0)
        return new
