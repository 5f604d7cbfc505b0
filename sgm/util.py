"""sgm.util names used by the hot path (reference: sgm/util.py:168-199)."""
from ccedit_amd.config import get_obj_from_str, instantiate_from_config  # noqa: F401
from ccedit_amd.sampling import append_dims, append_zero, default  # noqa: F401


def exists(x):
    return x is not None


def disabled_train(self, mode=True):
    return self
