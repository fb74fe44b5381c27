"""Hot-path modules (same names as reference nn/modules/__init__.py for the classes the DEAL-YOLO YAMLs use)."""
from .block import DFL, SPPF, Bottleneck, C2f
from .conv import Concat, Conv, LDConv, autopad
from .head import Detect

__all__ = ("Conv", "LDConv", "Concat", "DFL", "SPPF", "C2f", "Bottleneck", "Detect", "autopad")
