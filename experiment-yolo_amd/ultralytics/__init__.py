"""MI355X-native drop-in for the DEAL-YOLO hot path of adityaX1412/Experiment-YOLO (an Ultralytics-YOLOv8 fork).

Only the path scoped by SURVEY.md section 8 exists here; everything runs through libdealyolo_hip.so.
"""
__version__ = "8.1.9+dealyolo.hip.0"

from .engine.model import YOLO  # noqa: E402


def RTDETR(*a, **k):
    raise NotImplementedError("RTDETR is outside the DEAL-YOLO hot path (val.py of the reference uses it)")


__all__ = ("YOLO", "RTDETR", "__version__")
