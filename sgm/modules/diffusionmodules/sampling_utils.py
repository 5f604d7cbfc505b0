from ccedit_amd.sampling import (NoDynamicThresholding, get_ancestral_step, linear_multistep_coeff,  # noqa: F401
                                 to_neg_log_sigma, to_sigma)
