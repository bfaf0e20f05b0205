"""Overlay for the reference checkout: drop this file over `lib/model/DSTformer.py`
of Walter0807/MotionBERT (or put this repository first on PYTHONPATH) and every
reference entry point -- `lib/utils/learning.py:6,83-85` (load_backbone),
`train.py`, `train_action.py`, `infer_wild.py` -- constructs the MI355X-native
backbone instead of the PyTorch one, unchanged otherwise.  See INTEGRATION.md.
"""
from motionbert_amd.model import DSTformer, Block, Attention, MLP  # noqa: F401

__all__ = ['DSTformer', 'Block', 'Attention', 'MLP']
