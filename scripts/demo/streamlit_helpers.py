"""Only the piece of the reference's scripts/demo/streamlit_helpers.py that the sampling entry points reach
(`--sdedit_denoise_strength`, scripts/sampling/util.py:421-426): the SDEdit sigma-pruning wrapper."""
from ccedit_amd.sampling import Img2ImgDiscretizationWrapper  # noqa: F401
