from .block import Add, ScalSeq, Zoom_cat

__all__ = ("Add", "ScalSeq", "Zoom_cat")
