"""Pipeline helpers of the sampling entry points — counterpart of the reference's scripts/sampling/util.py for the
functions on the path (same names, arguments and defaults):

  create_model            util.py:38-42     yaml -> instantiate_from_config(config.model)
  init_sampling           util.py:385-425   (+ get_discretization :428-448, get_guider :451-480, get_sampler :483-556)
  prior_latent            sampling_tv2v.py:371-376, sampling_tv2v_ref.py:415-437   a*encode(keyframes/ref) + b*randn
  sdedit_start            sampling_tv2v.py:436-448   noised latent for --sdedit_denoise_strength
  save_frames / resume log                      sampling_tv2v.py:473-515 (numpy frames instead of mp4: no codec libs offline)

Video decoding, depth annotators, CLIP and the LoRA / base-model merge (util.py:45-272, 689-762) are outside this
build; conditioning arrives as tensors.
"""
from __future__ import annotations

import json
import os
from typing import Dict, Optional

import numpy as np
import torch

from ccedit_amd.config import instantiate_from_config, load_config

_DD = "sgm.modules.diffusionmodules."
SAMPLERS_BUILT = ("EulerAncestralSampler", "DPMPP2SAncestralSampler")


def create_model(config_path: str, device="cuda"):
    config = load_config(config_path)
    with torch.device(device):
        return instantiate_from_config(config.model)


def get_discretization(discretization: str) -> dict:
    if discretization == "LegacyDDPMDiscretization":
        return {"target": _DD + "discretizer.LegacyDDPMDiscretization"}
    raise NotImplementedError(f"discretization {discretization}: the shipped CCEdit commands use LegacyDDPMDiscretization")


def get_guider(guider_config_target=_DD + "guiders.VanillaCFG", scale=7.5) -> dict:
    return {"target": guider_config_target,
            "params": {"scale": scale, "dyn_thresh_config": {"target": _DD + "sampling_utils.NoDynamicThresholding"}}}


def get_sampler(sampler_name: str, steps: int, discretization_config: dict, guider_config: dict):
    if sampler_name not in SAMPLERS_BUILT:
        raise NotImplementedError(f"sampler {sampler_name}: built samplers are {SAMPLERS_BUILT}")
    return instantiate_from_config(dict(target=_DD + "sampling." + sampler_name, params=dict(
        num_steps=steps, discretization_config=discretization_config, guider_config=guider_config,
        eta=1.0, s_noise=1.0, verbose=True)))


def init_sampling(sample_steps=50, sampler_name="DPMPP2SAncestralSampler", discretization_name="LegacyDDPMDiscretization",
                  guider_config_target=_DD + "guiders.VanillaCFG", cfg_scale=7.5, img2img_strength=1.0):
    assert 1 <= sample_steps <= 1000, "sample_steps must be between 1 and 1000, but got {}".format(sample_steps)
    sampler = get_sampler(sampler_name, sample_steps, get_discretization(discretization_name),
                          get_guider(guider_config_target=guider_config_target, scale=cfg_scale))
    if img2img_strength < 1.0:
        from scripts.demo.streamlit_helpers import Img2ImgDiscretizationWrapper
        sampler.discretization = Img2ImgDiscretizationWrapper(sampler.discretization, strength=img2img_strength)
    return sampler


def prior_latent(model, randn: torch.Tensor, coeff_x: float, coeff_noise: float, keyframes: Optional[torch.Tensor] = None,
                 ref: Optional[torch.Tensor] = None, prior_type: str = "video") -> torch.Tensor:
    """randn <- coeff_x * prior + coeff_noise * randn, prior = encode_first_stage(keyframes) ['video'], of the
    reference image repeated over T ['ref'], or their sum ['video_ref']."""
    from ccedit_amd import ops
    assert 0.0 < coeff_x <= 1.0, "prior_coefficient_x should be in (0.0, 1.0], but got {}".format(coeff_x)
    t = randn.shape[2]
    if prior_type == "video":
        prior = model.encode_first_stage(keyframes)
    elif prior_type == "ref":
        prior = model.encode_first_stage(ref)[:, :, None].expand(-1, -1, t, -1, -1).contiguous()
    elif prior_type == "video_ref":
        pv = model.encode_first_stage(keyframes)
        pr = model.encode_first_stage(ref)[:, :, None].expand(-1, -1, t, -1, -1).contiguous()
        prior = ops.axpby(pv, pr, 1.0, 1.0)
    else:
        raise NotImplementedError
    return ops.axpby(prior.contiguous(), randn.float().contiguous(), coeff_x, coeff_noise)


def sdedit_start(model, sampler, keyframes: torch.Tensor) -> torch.Tensor:
    """z = encode(keyframes); noised_z = (z + randn_like(z) * sigma0) / sqrt(1 + sigma0^2) with sigma0 the first of the
    pruned schedule (hard-coded DDPM-like scaling, as in the reference)."""
    from ccedit_amd import ops
    z = model.encode_first_stage(keyframes)
    noise = torch.randn_like(z)
    sigmas = sampler.discretization(sampler.num_steps)
    s0 = float(sigmas[0])
    inv = 1.0 / float(torch.sqrt(1.0 + sigmas[0].float() ** 2.0))
    return ops.axpby(z.contiguous(), noise.contiguous(), inv, s0 * inv)


def load_conditioning(path: str) -> Dict[str, torch.Tensor]:
    if path.endswith(".safetensors"):
        from safetensors.torch import load_file
        return load_file(path)
    return torch.load(path, map_location="cpu")


def save_frames(save_path: str, tag: str, x: torch.Tensor) -> str:
    """x: decoded (1,3,T,H,W) in [-1,1] -> <save_path>/result/<tag>.npy with (T,H,W,3) in [0,1] (:473-475)."""
    os.makedirs(os.path.join(save_path, "result"), exist_ok=True)
    x = torch.clamp((x + 1.0) / 2.0, 0.0, 1.0)
    out = os.path.join(save_path, "result", tag + ".npy")
    np.save(out, x[0].permute(1, 2, 3, 0).cpu().numpy())
    return out


class ResumeLog:
    """log_info.json with the list of finished samples: re-running skips them unless --disable_check_repeat."""

    def __init__(self, save_path: str):
        os.makedirs(save_path, exist_ok=True)
        self.path = os.path.join(save_path, "log_info.json")
        self.log = json.load(open(self.path)) if os.path.exists(self.path) else {"done": []}

    def done(self, tag: str) -> bool:
        return tag in self.log["done"]

    def add(self, tag: str) -> None:
        self.log["done"].append(tag)
        json.dump(self.log, open(self.path, "w"))
