"""Test-side alias of the torch restatement of the kernel set (see oracle/torch_ops.py)."""
from oracle.torch_ops import MockOps  # noqa: F401
