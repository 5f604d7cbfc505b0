from ccedit_amd.sampling import (AncestralSampler, BaseDiffusionSampler, DPMPP2MSampler, DPMPP2SAncestralSampler,  # noqa: F401
                                 EDMSampler, EulerAncestralSampler, EulerEDMSampler, HeunEDMSampler,
                                 LinearMultistepSampler, SingleStepDiffusionSampler)
