from ccedit_amd.sampling import (Discretization, LegacyDDPMDiscretization,  # noqa: F401
                                 generate_roughly_equally_spaced_steps)
