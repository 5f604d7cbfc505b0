from ccedit_amd.vae import AutoencoderKL, AutoencoderKLInferenceWrapper  # noqa: F401
