"""Import path of the reference's ``YOLO`` class (models/yolo/model.py:12-40)."""
from ...engine.model import YOLO

__all__ = ("YOLO",)
