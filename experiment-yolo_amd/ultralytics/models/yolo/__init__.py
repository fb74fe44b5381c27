from . import detect

__all__ = ("detect",)
