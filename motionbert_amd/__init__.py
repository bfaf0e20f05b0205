"""MI355X-native DSTformer hot path (MotionBERT backbone) -- see DESIGN.md."""
from .model import DSTformer  # noqa: F401

__all__ = ['DSTformer']
__version__ = '0.1.0'
