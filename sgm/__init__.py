"""Drop-in `sgm` namespace: the dotted `target:` strings of CCEdit's inference configs
(configs/inference_ccedit/*.yaml) resolve to the MI355X-native classes in `ccedit_amd`.
Only the denoising hot path is provided; this is own code, not the reference package."""
from .util import instantiate_from_config  # noqa: F401
