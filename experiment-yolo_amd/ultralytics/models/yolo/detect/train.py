"""Import path of the reference's detection trainer (models/yolo/detect/train.py)."""
from ....engine.trainer import DetectionTrainer

__all__ = ("DetectionTrainer",)
