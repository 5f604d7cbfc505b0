from ccedit_amd.sampling import Denoiser, DiscreteDenoiser  # noqa: F401
