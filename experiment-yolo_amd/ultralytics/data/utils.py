"""Dataset descriptors and label files in YOLO format (drop-in for the detect-task subset of reference data/utils.py:
``img2label_paths`` :44-47, ``verify_image_label`` :96-165, ``check_det_dataset`` :252-343).  No download paths: a missing
dataset is an error here (the reference would try to fetch it)."""
from __future__ import annotations

import os
from pathlib import Path

import numpy as np
import yaml
from PIL import Image

IMG_FORMATS = "bmp", "dng", "jpeg", "jpg", "mpo", "png", "tif", "tiff", "webp", "pfm"  # data/utils.py:39


def img2label_paths(img_paths):
    """.../images/x.png -> .../labels/x.txt (last '/images/' occurrence)."""
    sa, sb = f"{os.sep}images{os.sep}", f"{os.sep}labels{os.sep}"
    return [sb.join(x.rsplit(sa, 1)).rsplit(".", 1)[0] + ".txt" for x in img_paths]


def exif_size(img: Image.Image):
    """PIL size corrected for EXIF orientation 6 / 8 (data/utils.py:61-73)."""
    s = img.size
    if img.format == "JPEG":
        try:
            rotation = (img.getexif() or {}).get(274, None)
            if rotation in (6, 8):
                s = s[1], s[0]
        except Exception:
            pass
    return s


class _Reject(Exception):
    """A pair the reference would drop as corrupt."""


def _read_labels(lb_file, num_cls):
    """Label text -> (n,5) float32 rows [cls, x, y, w, h] and a note; raises _Reject on the reference's rejection rules."""
    rows = [ln.split() for ln in Path(lb_file).read_text().strip().splitlines() if ln]
    if not rows:
        return np.zeros((0, 5), dtype=np.float32), ""
    if max(len(r) for r in rows) > 6:
        raise _Reject("segment labels are not on the detect path")
    try:
        lb = np.array(rows, dtype=np.float32)
    except ValueError as e:
        raise _Reject(str(e))
    if lb.ndim != 2 or lb.shape[1] != 5:
        raise _Reject(f"labels require 5 columns, {lb.shape[-1] if lb.ndim == 2 else 'ragged'} columns detected")
    if lb[:, 1:].max() > 1:
        raise _Reject(f"non-normalized or out of bounds coordinates {lb[:, 1:][lb[:, 1:] > 1]}")
    if lb.min() < 0:
        raise _Reject(f"negative label values {lb[lb < 0]}")
    if lb[:, 0].max() > num_cls:
        raise _Reject(f"Label class {int(lb[:, 0].max())} exceeds dataset class count {num_cls}")
    first = np.unique(lb, axis=0, return_index=True)[1]  # first occurrence of every distinct row, rows come back SORTED
    note = f"{len(lb) - len(first)} duplicate labels removed" if len(first) < len(lb) else ""
    return (lb[first] if note else lb), note


def verify_image_label(im_file, lb_file, num_cls):
    """One image/label pair -> (im_file | None, labels (n,5) float32 [cls, x, y, w, h], (h, w), n_missing, n_found, n_empty,
    n_corrupt, message).  Same acceptance rules as the reference (data/utils.py:96-165): image at least 10x10 and of a known
    format; five columns; coordinates normalised (<= 1) and non-negative; class ids within the dataset's count; duplicate rows
    removed (np.unique, i.e. the kept rows come back sorted); any violation drops the whole image as corrupt."""
    missing = found = empty = 0
    try:
        with Image.open(im_file) as im:
            im.verify()
            w, h = exif_size(im)
            fmt = (im.format or "").lower()
        if min(h, w) < 10:
            raise _Reject(f"image size {(h, w)} <10 pixels")
        if fmt not in IMG_FORMATS:
            raise _Reject(f"invalid image format {fmt}")
        note = ""
        if os.path.isfile(lb_file):
            found = 1
            lb, note = _read_labels(lb_file, num_cls)
            empty = int(len(lb) == 0)
        else:
            missing, lb = 1, np.zeros((0, 5), dtype=np.float32)
        return im_file, lb, (h, w), missing, found, empty, 0, f"WARNING {im_file}: {note}" if note else ""
    except Exception as e:  # noqa: BLE001 -- the reference catches everything here too
        return None, None, None, missing, found, empty, 1, f"WARNING {im_file}: ignoring corrupt image/label: {e}"


def check_det_dataset(dataset):
    """Dataset YAML -> dict with absolute 'path', 'train', 'val' (, 'test'), 'nc', 'names' (index -> name)."""
    file = Path(dataset)
    if not file.is_file():
        raise FileNotFoundError(f"'{dataset}' does not exist")
    with open(file, errors="ignore", encoding="utf-8") as f:
        data = yaml.safe_load(f) or {}
    data["yaml_file"] = str(file)
    for k in "train", "val":
        if k not in data:
            if k != "val" or "validation" not in data:
                raise SyntaxError(f"{dataset} '{k}:' key missing.\n'train' and 'val' are required in all data YAMLs.")
            data["val"] = data.pop("validation")
    if "names" not in data and "nc" not in data:
        raise SyntaxError(f"{dataset} key missing.\n either 'names' or 'nc' are required in all data YAMLs.")
    if "names" in data and "nc" in data and len(data["names"]) != data["nc"]:
        raise SyntaxError(f"{dataset} 'names' length {len(data['names'])} and 'nc: {data['nc']}' must match.")
    if "names" not in data:
        data["names"] = [f"class_{i}" for i in range(data["nc"])]
    else:
        data["nc"] = len(data["names"])
    if isinstance(data["names"], list):
        data["names"] = dict(enumerate(data["names"]))
    data["names"] = {int(k): str(v) for k, v in data["names"].items()}
    path = Path(data.get("path") or file.parent)
    if not path.is_absolute():
        path = (file.parent / path).resolve()  # the reference resolves against its global datasets dir; here: the YAML's folder
    data["path"] = path
    for k in "train", "val", "test":
        if data.get(k):
            if isinstance(data[k], str):
                x = (path / data[k]).resolve()
                if not x.exists() and data[k].startswith("../"):
                    x = (path / data[k][3:]).resolve()
                data[k] = str(x)
            else:
                data[k] = [str((path / x).resolve()) for x in data[k]]
    val = data["val"] if isinstance(data["val"], list) else [data["val"]]
    missing = [x for x in val if not Path(x).exists()]
    if missing:
        raise FileNotFoundError(f"Dataset '{dataset}' images not found, missing path '{missing[0]}' (no download path here)")
    return data
