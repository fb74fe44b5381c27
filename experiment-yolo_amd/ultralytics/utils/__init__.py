"""Host utilities the hot path needs (LOGGER, rank environment, default cfg)."""
import logging
import os

from ..cfg import DEFAULT_CFG, DEFAULT_CFG_DICT, IterableSimpleNamespace  # noqa: F401

RANK = int(os.getenv("RANK", -1))
LOCAL_RANK = int(os.getenv("LOCAL_RANK", -1))
LOGGER = logging.getLogger("ultralytics")
if not LOGGER.handlers:
    _h = logging.StreamHandler()
    _h.setFormatter(logging.Formatter("%(message)s"))
    LOGGER.addHandler(_h)
    LOGGER.setLevel(logging.INFO if RANK in (-1, 0) else logging.ERROR)
