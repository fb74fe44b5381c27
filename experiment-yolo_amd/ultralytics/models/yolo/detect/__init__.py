from .val import DetectionValidator

__all__ = ("DetectionValidator",)
