"""Import path of the reference's detection predictor (models/yolo/detect/predict.py)."""
from ....engine.predictor import DetectionPredictor

__all__ = ("DetectionPredictor",)
