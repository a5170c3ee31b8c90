"""Drop-in for the reference's `distributed.py` (train.py:32-34 imports these three names)."""
from flowtron_amd.dist import (FlatArena, apply_gradient_allreduce, init_distributed,  # noqa: F401
                               reduce_tensor, reduce_tensors)
