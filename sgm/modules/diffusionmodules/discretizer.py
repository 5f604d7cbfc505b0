from ccedit_amd.sampling import (Discretization, EDMDiscretization, LegacyDDPMDiscretization,  # noqa: F401
                                 generate_roughly_equally_spaced_steps)
