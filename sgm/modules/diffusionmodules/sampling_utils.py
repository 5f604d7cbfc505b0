from ccedit_amd.sampling import NoDynamicThresholding, get_ancestral_step, to_neg_log_sigma, to_sigma  # noqa: F401
