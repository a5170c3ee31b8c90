"""Drop-in for the reference's `radam.py` (train.py:30 `from radam import RAdam`): fused HIP implementation."""
from flowtron_amd.optim import RAdam  # noqa: F401
