from ccedit_amd.sampling import (AncestralSampler, BaseDiffusionSampler, DPMPP2SAncestralSampler,  # noqa: F401
                                 EulerAncestralSampler, SingleStepDiffusionSampler)
