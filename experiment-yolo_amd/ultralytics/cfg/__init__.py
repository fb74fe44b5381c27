"""Configuration (drop-in for the parts of reference cfg/__init__.py the hot path uses): typed merge of default.yaml
with overrides; unknown keys are rejected like ``check_dict_alignment`` does (reference cfg/__init__.py:286)."""
from __future__ import annotations

from pathlib import Path
from types import SimpleNamespace

import yaml

DEFAULT_CFG_PATH = Path(__file__).resolve().parent / "default.yaml"
DEFAULT_CFG_DICT = yaml.safe_load(DEFAULT_CFG_PATH.read_text())
for _k, _v in DEFAULT_CFG_DICT.items():
    if isinstance(_v, str) and _v.lower() == "none":
        DEFAULT_CFG_DICT[_k] = None

CFG_FLOAT_KEYS = "warmup_epochs", "box", "cls", "dfl", "degrees", "shear", "time", "loss_scale"
CFG_FRACTION_KEYS = ("dropout", "iou", "lr0", "lrf", "momentum", "weight_decay", "warmup_momentum", "warmup_bias_lr",
                     "label_smoothing", "hsv_h", "hsv_s", "hsv_v", "translate", "scale", "perspective", "flipud", "fliplr",
                     "mosaic", "mixup", "copy_paste", "conf", "fraction", "iou_ratio")
CFG_INT_KEYS = ("epochs", "patience", "batch", "workers", "seed", "close_mosaic", "max_det", "vid_stride", "nbs", "save_period", "mask_ratio",
                "nmax")
CFG_BOOL_KEYS = ("save", "exist_ok", "verbose", "deterministic", "single_cls", "rect", "cos_lr", "amp", "val", "half",
                 "agnostic_nms", "plots", "wiou", "nwd", "hipgraph", "multi_scale", "overlap_mask")


class IterableSimpleNamespace(SimpleNamespace):
    def __iter__(self):
        return iter(vars(self).items())

    def get(self, key, default=None):
        return getattr(self, key, default)


def get_cfg(cfg=DEFAULT_CFG_DICT, overrides=None):
    """Merge ``cfg`` (dict | namespace | yaml path) with ``overrides``; type-check like reference cfg/__init__.py:192-248."""
    if isinstance(cfg, (str, Path)):
        cfg = yaml.safe_load(Path(cfg).read_text())
    elif isinstance(cfg, SimpleNamespace):
        cfg = vars(cfg)
    cfg = dict(cfg)
    if overrides:
        overrides = dict(overrides)
        bad = [k for k in overrides if k not in DEFAULT_CFG_DICT]
        if bad:
            raise SyntaxError(f"'{bad[0]}' is not a valid YOLO argument (valid: see ultralytics/cfg/default.yaml)")
        cfg.update(overrides)
    for k, v in cfg.items():
        if v is None:
            continue
        if k in CFG_FLOAT_KEYS and not isinstance(v, (int, float)):
            raise TypeError(f"'{k}={v}' is of invalid type {type(v).__name__}; must be int or float")
        if k in CFG_FRACTION_KEYS:
            if not isinstance(v, (int, float)):
                raise TypeError(f"'{k}={v}' is of invalid type {type(v).__name__}; must be int or float")
            if not 0.0 <= v <= 1.0:
                raise ValueError(f"'{k}={v}' is an invalid value; must be between 0.0 and 1.0")
        if k in CFG_INT_KEYS and not isinstance(v, int):
            raise TypeError(f"'{k}={v}' is of invalid type {type(v).__name__}; must be an int")
        if k in CFG_BOOL_KEYS and not isinstance(v, bool):
            raise TypeError(f"'{k}={v}' is of invalid type {type(v).__name__}; must be a bool")
    return IterableSimpleNamespace(**cfg)


DEFAULT_CFG = get_cfg(DEFAULT_CFG_DICT)
