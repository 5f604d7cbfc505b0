"""Sampling numerics of the hot path: sigma schedule, DiscreteDenoiser (eps-prediction scaling),
classifier-free guidance and the DPM-Solver++(2S) ancestral sampler.

Mirrors the reference classes one-for-one (same names, constructor arguments, call signatures):
  sgm/modules/diffusionmodules/discretizer.py   Discretization, LegacyDDPMDiscretization
  sgm/modules/diffusionmodules/denoiser*.py     Denoiser, DiscreteDenoiser, EpsScaling, EpsWeighting
  sgm/modules/diffusionmodules/guiders.py       VanillaCFG, VanillaCFGTV2V, IdentityGuider
  sgm/modules/diffusionmodules/sampling.py      BaseDiffusionSampler ... DPMPP2SAncestralSampler
  sgm/modules/diffusionmodules/sampling_utils.py get_ancestral_step, to_d, to_(neg_log_)sigma

Design difference: every sigma-derived quantity (schedule, ancestral split, exponential-integrator
multipliers, table quantisation, timestep indices) is a handful of fp32 scalars, so it is evaluated
on the HOST with the reference's exact sequence of fp32 operations (1-element CPU tensors) — the
schedule and index tensors are bit-exact by construction and the per-step device->host sync of the
reference (`torch.sum(sigma_down) < 1e-14`, sampling.py:390) disappears.  Arithmetic on the latent
(417,792 fp32 elements at 17x64x96) runs in the HIP kernels `ccedit_axpby` / `ccedit_cfg_denoise`.
"""
from __future__ import annotations

from functools import partial
from typing import Callable, Dict, Optional, Union

import numpy as np
import torch
import torch.nn as nn

from . import ops
from .config import instantiate_from_config

DEFAULT_GUIDER = {"target": "sgm.modules.diffusionmodules.guiders.IdentityGuider"}


def default(val, d):
    return val if val is not None else (d() if callable(d) and not isinstance(d, dict) else d)


def append_dims(x: torch.Tensor, target_dims: int) -> torch.Tensor:
    """sgm/util.py:192-199"""
    extra = target_dims - x.ndim
    if extra < 0:
        raise ValueError(f"input has {x.ndim} dims but target_dims is {target_dims}, which is less")
    return x[(...,) + (None,) * extra]


def append_zero(x: torch.Tensor) -> torch.Tensor:
    return torch.cat([x, x.new_zeros([1])])


# ------------------------------------------------------------------------------------------
# discretisation
# ------------------------------------------------------------------------------------------
def make_beta_schedule(schedule, n_timestep, linear_start=1e-4, linear_end=2e-2):
    """diffusionmodules/util.py:24-37 ("linear" = scaled-linear in sqrt space, float64)."""
    if schedule != "linear":
        raise NotImplementedError(schedule)
    betas = torch.linspace(linear_start ** 0.5, linear_end ** 0.5, n_timestep, dtype=torch.float64, device="cpu") ** 2
    return betas.numpy()


def generate_roughly_equally_spaced_steps(num_substeps: int, max_step: int) -> np.ndarray:
    return np.linspace(max_step - 1, 0, num_substeps, endpoint=False).astype(int)[::-1]


class Discretization:
    def __call__(self, n, do_append_zero=True, device="cpu", flip=False):
        sigmas = self.get_sigmas(n, device=device)
        sigmas = append_zero(sigmas) if do_append_zero else sigmas
        return sigmas if not flip else torch.flip(sigmas, (0,))

    def get_sigmas(self, n, device):
        raise NotImplementedError


class LegacyDDPMDiscretization(Discretization):
    """discretizer.py:42-69.  sigma = ((1-abar)/abar) cast to f32, THEN sqrt; descending order."""

    def __init__(self, linear_start=0.00085, linear_end=0.0120, num_timesteps=1000):
        self.num_timesteps = num_timesteps
        betas = make_beta_schedule("linear", num_timesteps, linear_start=linear_start, linear_end=linear_end)
        self.alphas_cumprod = np.cumprod(1.0 - betas, axis=0)

    def get_sigmas(self, n, device="cpu"):
        if n < self.num_timesteps:
            timesteps = generate_roughly_equally_spaced_steps(n, self.num_timesteps)
            alphas_cumprod = self.alphas_cumprod[timesteps]
        elif n == self.num_timesteps:
            alphas_cumprod = self.alphas_cumprod
        else:
            raise ValueError
        sigmas = torch.tensor((1 - alphas_cumprod) / alphas_cumprod, dtype=torch.float32, device="cpu") ** 0.5   # host math
        return torch.flip(sigmas, (0,)).to(device)


# ------------------------------------------------------------------------------------------
# denoiser
# ------------------------------------------------------------------------------------------
class EDMDiscretization(Discretization):
    """discretizer.py:28-39 (Karras rho schedule)."""

    def __init__(self, sigma_min=0.02, sigma_max=80.0, rho=7.0):
        self.sigma_min, self.sigma_max, self.rho = sigma_min, sigma_max, rho

    def get_sigmas(self, n, device="cpu"):
        ramp = torch.linspace(0, 1, n, device="cpu")                      # host math
        min_inv_rho = self.sigma_min ** (1 / self.rho)
        max_inv_rho = self.sigma_max ** (1 / self.rho)
        return ((max_inv_rho + ramp * (min_inv_rho - max_inv_rho)) ** self.rho).to(device)


class Img2ImgDiscretizationWrapper:
    """scripts/demo/streamlit_helpers.py:212-233 (SDEdit): wraps a discretizer and keeps only the
    max(int(strength * len), 1) smallest sigmas (the tail of the descending schedule, trailing zero included)."""

    def __init__(self, discretization, strength: float = 1.0):
        self.discretization = discretization
        self.strength = strength
        assert 0.0 <= self.strength <= 1.0

    def __call__(self, *args, **kwargs):
        full = self.discretization(*args, **kwargs)                 # descending, trailing zero included
        keep = max(int(self.strength * len(full)), 1)
        return full[len(full) - keep:]                              # the `keep` smallest sigmas, order unchanged


class EpsWeighting:
    def __call__(self, sigma):
        return sigma ** -2.0


class EpsScaling:
    """denoiser_scaling.py:16-22"""

    def __call__(self, sigma):
        c_skip = torch.ones_like(sigma, device=sigma.device)
        c_out = -sigma
        c_in = 1 / (sigma ** 2 + 1.0) ** 0.5
        c_noise = sigma.clone()
        return c_skip, c_out, c_in, c_noise


class Denoiser(nn.Module):
    """denoiser.py:6-40.  `sigma` may live on any device; the scalar math runs on the host."""

    def __init__(self, weighting_config, scaling_config):
        super().__init__()
        self.weighting = instantiate_from_config(weighting_config)
        self.scaling = instantiate_from_config(scaling_config)

    def possibly_quantize_sigma(self, sigma):
        return sigma

    def possibly_quantize_c_noise(self, c_noise):
        return c_noise

    def w(self, sigma):
        return self.weighting(sigma)

    def __call__(self, network, input, sigma, cond):
        sigma = self.possibly_quantize_sigma(sigma.detach().to("cpu", torch.float32))
        c_skip, c_out, c_in, c_noise = self.scaling(sigma)
        # int64 table indices (discrete) or sigmas.  Through pinned memory and without blocking: a plain `.to(device)` of a pageable host
        # tensor waits for everything queued on the stream — one full host sync per evaluation, the host could never enqueue the next
        # evaluation's replay under the running one
        c_noise_q = self.possibly_quantize_c_noise(c_noise)
        c_noise_dev = c_noise_q.pin_memory().to(input.device, non_blocking=True) if input.is_cuda else c_noise_q.to(input.device)
        b = input.shape[0]
        x = input.float().contiguous()
        xs = torch.empty_like(x)
        for i in range(b):                                    # input * c_in
            ops.axpby(x[i], x[i], float(c_in[i]), 0.0, out=xs[i])
        # The guider marks a batch whose two halves are the SAME latent (VanillaCFG.prepare_inputs); with equal sigmas the scaled
        # input and the timestep indices are twins too.  The mark travels as a Python attribute on the tensors handed to the
        # network, which then shares the halves' common prefix without comparing them on the device (a host sync per evaluation
        # would expose the launch of every replayed graph: measured +4 ms per evaluation of a clip).
        k = b // 2
        if (ops.get_mark(input, "_cfg_twin_halves") is True and b % 2 == 0 and all(float(c_in[i]) == float(c_in[i + k]) for i in range(k))
                and bool(torch.equal(c_noise[:k], c_noise[k:]))):
            ops.set_mark(xs, "_cfg_twin_halves", True)         # (mark + in-place version: a later in-place edit voids it, ops.get_mark)
            ops.set_mark(c_noise_dev, "_cfg_twin_halves", True)
        net = network(xs, c_noise_dev, cond).float().contiguous()
        out = torch.empty_like(x)
        for i in range(b):                                    # net * c_out + input * c_skip
            ops.axpby(net[i], x[i], float(c_out[i]), float(c_skip[i]), out=out[i])
        return out


class DiscreteDenoiser(Denoiser):
    """denoiser.py:43-75: sigma and c_noise are quantised to the 1000-entry table by argmin."""

    def __init__(self, weighting_config, scaling_config, num_idx, discretization_config, do_append_zero=False,
                 quantize_c_noise=True, flip=True):
        super().__init__(weighting_config, scaling_config)
        sigmas = instantiate_from_config(discretization_config)(num_idx, do_append_zero=do_append_zero, flip=flip)
        self.register_buffer("sigmas", sigmas)
        self._host_sigmas = sigmas.detach().to("cpu", torch.float32).clone()
        self.quantize_c_noise = quantize_c_noise

    def _table(self, like: torch.Tensor) -> torch.Tensor:
        return self._host_sigmas if like.device.type == "cpu" else self.sigmas.to(like.device)

    def sigma_to_idx(self, sigma):
        dists = sigma - self._table(sigma)[:, None]
        return dists.abs().argmin(dim=0).view(sigma.shape)

    def idx_to_sigma(self, idx):
        return self._table(idx)[idx]

    def possibly_quantize_sigma(self, sigma):
        return self.idx_to_sigma(self.sigma_to_idx(sigma))

    def possibly_quantize_c_noise(self, c_noise):
        return self.sigma_to_idx(c_noise) if self.quantize_c_noise else c_noise


# ------------------------------------------------------------------------------------------
# guiders
# ------------------------------------------------------------------------------------------
class NoDynamicThresholding:
    def __call__(self, uncond, cond, scale):
        # u + scale * (c - u) on the device: y = (1-scale)*u + scale*c  (sampling_utils.py:7-9)
        u, c = uncond.float().contiguous(), cond.float().contiguous()
        return ops.axpby(u, c, 1.0 - float(scale), float(scale))


class VanillaCFG:
    """guiders.py:8-40"""

    _CAT_KEYS = ("vector", "crossattn", "concat", "cond_feat")

    def __init__(self, scale, dyn_thresh_config=None):
        self.scale = scale
        self._cat_cache = {}
        self.scale_schedule = lambda sigma: scale
        self.dyn_thresh = instantiate_from_config(default(dyn_thresh_config, {
            "target": "sgm.modules.diffusionmodules.sampling_utils.NoDynamicThresholding"}))

    def __call__(self, x, sigma):
        x_u, x_c = x.chunk(2)
        return self.dyn_thresh(x_u, x_c, self.scale_schedule(sigma))

    def prepare_inputs(self, x, s, c, uc):
        c_out = dict()
        for k in c:
            if k in self._CAT_KEYS:
                # uc FIRST (guiders.py:63).  The conditioning tensors do not change during a clip: the
                # concatenation (160 MB for the 17x512x768 hint) is built once per (uc[k], c[k]) pair.  The entry
                # HOLDS the two source tensors and is matched by object identity (+ in-place version), so a later
                # clip whose tensors land on a recycled address can never hit it.
                hit = self._cat_cache.get(k)
                if hit is None or hit[0] is not uc[k] or hit[1] is not c[k] or hit[2] != (uc[k]._version, c[k]._version):
                    hit = (uc[k], c[k], (uc[k]._version, c[k]._version), torch.cat((uc[k], c[k]), 0))
                    # are the two halves the same values (control_hint, cond_feat: the scripts give uc a clone of c's)?  Compared once per
                    # (uc[k], c[k]) pair, remembered on the concatenation for the network's shared CFG prefix
                    ops.set_mark(hit[3], "_halves_equal", bool(uc[k].shape == c[k].shape and torch.equal(uc[k], c[k])))
                    self._cat_cache[k] = hit
                c_out[k] = hit[3]
            else:
                assert c[k] == uc[k]
                c_out[k] = c[k]
        x2, s2 = torch.cat([x] * 2), torch.cat([s] * 2)
        ops.set_mark(x2, "_cfg_twin_halves", True)          # the same latent twice (see Denoiser.__call__)
        return x2, s2, c_out


class VanillaCFGTV2V(VanillaCFG):
    """guiders.py:56-67: control_hint (and interpolate_*) are batch-doubled too."""

    _CAT_KEYS = ("vector", "crossattn", "concat", "cond_feat", "control_hint", "interpolate_first", "interpolate_last",
                 "interpolate_first_last")


class IdentityGuider:
    def __call__(self, x, sigma):
        return x

    def prepare_inputs(self, x, s, c, uc):
        return x, s, {k: c[k] for k in c}


# ------------------------------------------------------------------------------------------
# samplers
# ------------------------------------------------------------------------------------------
def get_ancestral_step(sigma_from, sigma_to, eta=1.0):
    """sampling_utils.py:27-36"""
    if not eta:
        return sigma_to, 0.0
    sigma_up = torch.minimum(sigma_to, eta * (sigma_to ** 2 * (sigma_from ** 2 - sigma_to ** 2) / sigma_from ** 2) ** 0.5)
    sigma_down = (sigma_to ** 2 - sigma_up ** 2) ** 0.5
    return sigma_down, sigma_up


def to_neg_log_sigma(sigma):
    return sigma.log().neg()


def to_sigma(neg_log_sigma):
    return neg_log_sigma.neg().exp()


class BaseDiffusionSampler:
    """sampling.py:24-77.  `device` names where the latent lives; sigma scalars stay on the host."""

    def __init__(self, discretization_config, num_steps=None, guider_config=None, verbose=False, device="cuda"):
        self.num_steps = num_steps
        self.discretization = instantiate_from_config(discretization_config)
        self.guider = instantiate_from_config(default(guider_config, DEFAULT_GUIDER))
        self.verbose = verbose
        self.device = device

    def prepare_sampling_loop(self, x, cond, uc=None, num_steps=None):
        sigmas = self.discretization(self.num_steps if num_steps is None else num_steps, device="cpu")
        uc = default(uc, cond)
        # x *= sqrt(1 + sigma_0^2), in place like the reference (sampling.py:50)
        k = float(torch.sqrt(1.0 + sigmas[0] ** 2.0))
        xf = x if (x.dtype == torch.float32 and x.is_contiguous()) else x.float().contiguous()
        ops.axpby(xf, xf, k, 0.0, out=xf)
        num_sigmas = len(sigmas)
        s_in = torch.ones([x.shape[0]], dtype=torch.float32, device="cpu")          # host
        return xf, s_in, sigmas, num_sigmas, cond, uc

    def denoise(self, x, denoiser, sigma, cond, uc):
        denoised = denoiser(*self.guider.prepare_inputs(x, sigma, cond, uc))
        return self.guider(denoised, sigma)

    def get_sigma_gen(self, num_sigmas):
        """Step indices 0 .. num_sigmas - 2 (name kept: reference scripts call it); verbose wraps them in a progress bar."""
        steps = range(num_sigmas - 1)
        if not self.verbose:
            return steps
        what = f"{type(self).__name__} / {type(self.discretization).__name__} / {type(self.guider).__name__}"
        try:
            from tqdm import tqdm
        except ImportError:      # pragma: no cover
            print(f"[sampler] {what}: {num_sigmas - 1} steps")
            return steps
        return tqdm(steps, total=num_sigmas - 1, desc=what)

    def _walk(self, denoiser, x, cond, uc, num_steps, before_step=None, first_step=0, step_kwargs=None):
        """The loop every sampler entry point shares: (sigma_i, sigma_{i+1}) pairs of the schedule into `sampler_step`.
        before_step(x, i, sigmas) -> x edits the latent ahead of a step (inpainting / blending variants); first_step(num_sigmas)
        skips the head of the schedule (SDEdit); step_kwargs(i, sigmas, num_sigmas) adds per-step arguments (EDM churn)."""
        x, s_in, sigmas, num_sigmas, cond, uc = self.prepare_sampling_loop(x, cond, uc, num_steps)
        start = first_step(num_sigmas) if callable(first_step) else first_step
        for i in self.get_sigma_gen(num_sigmas):
            if i < start:
                continue
            if before_step is not None:
                x = before_step(x, i, sigmas)
            extra = step_kwargs(i, sigmas, num_sigmas) if step_kwargs is not None else {}
            x = self.sampler_step(s_in * sigmas[i], s_in * sigmas[i + 1], denoiser, x, cond, uc, **extra)
        return x


class SingleStepDiffusionSampler(BaseDiffusionSampler):
    def sampler_step(self, sigma, next_sigma, denoiser, x, cond, uc, *args, **kwargs):
        raise NotImplementedError

    def euler_step(self, x, d, dt):
        return x + dt * d

    def _noised_original(self, x0, sigma_i):
        """(x0 + randn * sigma_i) / sqrt(1 + sigma_i^2): the known content at the current noise level
        (sampling.py:150-152, 213-215, 236-238)."""
        noise = self.noise_sampler(x0).float().contiguous()
        inv = 1.0 / float(torch.sqrt(1.0 + sigma_i ** 2))
        return ops.axpby(x0.float().contiguous(), noise, inv, float(sigma_i) * inv)

    def _inpaint_blend(self, x, x0, mask, sigma_i):
        return ops.mask_blend(x, self._noised_original(x0, sigma_i), mask.float().expand_as(x).contiguous())


class AncestralSampler(SingleStepDiffusionSampler):
    """sampling.py:168-205"""

    def __init__(self, eta=1.0, s_noise=1.0, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.eta = eta
        self.s_noise = s_noise
        self.noise_sampler = lambda x: torch.randn_like(x)

    def ancestral_euler_step(self, x, denoised, sigma, sigma_down):
        # x + (x - denoised)/sigma * (sigma_down - sigma) == (1 + r) x - r denoised,  r = (sigma_down - sigma)/sigma
        r = float(((sigma_down - sigma) / sigma)[0])
        return ops.axpby(x, denoised, 1.0 + r, -r)

    def ancestral_step(self, x, sigma, next_sigma, sigma_up):
        noise = self.noise_sampler(x)              # drawn on every step, including the last (sampling.py:182-188)
        if float(next_sigma[0]) > 0.0:
            return ops.axpby(x, noise.float().contiguous(), 1.0, self.s_noise * float(sigma_up[0]))
        return x

    def __call__(self, denoiser, x, cond, uc=None, num_steps=None):
        return self._walk(denoiser, x, cond, uc, num_steps)

    def sample_inpainting(self, denoiser, x, cond, x0, mask, uc=None, num_steps=None):
        """sampling.py:206-225: before every step the region with mask == 0 is reset to the noised original."""
        return self._walk(denoiser, x, cond, uc, num_steps, before_step=lambda x, i, sig: self._inpaint_blend(x, x0, mask, sig[i]))

    def sampling_blending(self, denoiser, x, cond, x0, uc=None, num_steps=None):
        """sampling.py:227-249: the first T//2 frames are overwritten with the noised LAST T//2 frames of x0."""
        def overwrite_head(x, i, sig):
            half = x.shape[2] // 2
            x[:, :, :half] = self._noised_original(x0, sig[i])[:, :, half + 1:]
            return x
        return self._walk(denoiser, x, cond, uc, num_steps, before_step=overwrite_head)

    def sdedit(self, denoise_steps, denoiser, x, cond, uc=None, num_steps=None):
        """sampling.py:251-266: run only the last `denoise_steps` steps of the schedule."""
        return self._walk(denoiser, x, cond, uc, num_steps, first_step=lambda n: n - 1 - denoise_steps)


class EulerAncestralSampler(AncestralSampler):
    def sampler_step(self, sigma, next_sigma, denoiser, x, cond, uc):
        sigma_down, sigma_up = get_ancestral_step(sigma, next_sigma, eta=self.eta)
        denoised = self.denoise(x, denoiser, sigma, cond, uc)
        x = self.ancestral_euler_step(x, denoised, sigma, sigma_down)
        return self.ancestral_step(x, sigma, next_sigma, sigma_up)


class DPMPP2SAncestralSampler(AncestralSampler):
    """sampling.py:370-407.  All batch entries share one sigma (s_in * sigma_i), as in the reference's
    only call site; the per-step scalars are computed on the host in the reference's op order."""

    def get_variables(self, sigma, sigma_down):
        t, t_next = [to_neg_log_sigma(s) for s in (sigma, sigma_down)]
        h = t_next - t
        s = t + 0.5 * h
        return h, s, t, t_next

    def get_mult(self, h, s, t, t_next):
        mult1 = to_sigma(s) / to_sigma(t)
        mult2 = (-0.5 * h).expm1()
        mult3 = to_sigma(t_next) / to_sigma(t)
        mult4 = (-h).expm1()
        return mult1, mult2, mult3, mult4

    def sampler_step(self, sigma, next_sigma, denoiser, x, cond, uc=None, **kwargs):
        sigma_down, sigma_up = get_ancestral_step(sigma, next_sigma, eta=self.eta)
        denoised = self.denoise(x, denoiser, sigma, cond, uc)
        if torch.sum(sigma_down) < 1e-14:
            # save a network evaluation if all noise levels are 0 (host check, no device sync)
            x = self.ancestral_euler_step(x, denoised, sigma, sigma_down)
        else:
            h, s, t, t_next = self.get_variables(sigma, sigma_down)
            m1, m2, m3, m4 = [float(m[0]) for m in self.get_mult(h, s, t, t_next)]
            x2 = ops.axpby(x, denoised, m1, -m2)
            denoised2 = self.denoise(x2, denoiser, to_sigma(s), cond, uc)
            # sigma_down > 0 here, so the reference's torch.where picks x_dpmpp2s
            x = ops.axpby(x, denoised2, m3, -m4)
        return self.ancestral_step(x, sigma, next_sigma, sigma_up)


class EDMSampler(SingleStepDiffusionSampler):
    """sampling.py:86-133: Euler step on the probability-flow ODE with optional churn (gamma)."""

    def __init__(self, s_churn=0.0, s_tmin=0.0, s_tmax=float("inf"), s_noise=1.0, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.s_churn, self.s_tmin, self.s_tmax, self.s_noise = s_churn, s_tmin, s_tmax, s_noise
        self.noise_sampler = lambda x: torch.randn_like(x)

    def sampler_step(self, sigma, next_sigma, denoiser, x, cond, uc=None, gamma=0.0):
        sigma_hat = sigma * (gamma + 1.0)
        if gamma > 0:
            eps = self.noise_sampler(x).float().contiguous()
            x = ops.axpby(x, eps, 1.0, self.s_noise * float(((sigma_hat ** 2 - sigma ** 2) ** 0.5)[0]))
        denoised = self.denoise(x, denoiser, sigma_hat, cond, uc)
        dt = next_sigma - sigma_hat
        r = float((dt / sigma_hat)[0])
        euler_step = ops.axpby(x, denoised, 1.0 + r, -r)              # x + dt * (x - denoised) / sigma_hat
        return self.possible_correction_step(euler_step, x, denoised, sigma_hat, dt, next_sigma, denoiser, cond, uc)

    def possible_correction_step(self, euler_step, x, denoised, sigma_hat, dt, next_sigma, denoiser, cond, uc):
        raise NotImplementedError

    def _churn(self, i, sigmas, num_sigmas):
        """gamma of step i: s_churn spread over the steps (capped at sqrt(2) - 1) while sigma_i lies in [s_tmin, s_tmax], else 0."""
        inside = self.s_tmin <= sigmas[i] <= self.s_tmax
        return dict(gamma=min(self.s_churn / (num_sigmas - 1), 2 ** 0.5 - 1) if inside else 0.0)

    def __call__(self, denoiser, x, cond, uc=None, num_steps=None):
        return self._walk(denoiser, x, cond, uc, num_steps, step_kwargs=self._churn)

    def sample_inpainting(self, denoiser, x, cond, x0, mask, uc=None, num_steps=None):
        """sampling.py:138-166."""
        return self._walk(denoiser, x, cond, uc, num_steps, step_kwargs=self._churn,
                          before_step=lambda x, i, sig: self._inpaint_blend(x, x0, mask, sig[i]))


class EulerEDMSampler(EDMSampler):
    """sampling.py:307-311 — the reference script's default `--sampler_name`."""

    def possible_correction_step(self, euler_step, x, denoised, sigma_hat, dt, next_sigma, denoiser, cond, uc):
        return euler_step


class HeunEDMSampler(EDMSampler):
    """sampling.py:314-329: second evaluation at next_sigma, trapezoidal slope; plain Euler on the final step."""

    def possible_correction_step(self, euler_step, x, denoised, sigma_hat, dt, next_sigma, denoiser, cond, uc):
        if torch.sum(next_sigma) < 1e-14:
            return euler_step
        denoised2 = self.denoise(euler_step, denoiser, next_sigma, cond, uc)
        a = float((dt / (2.0 * sigma_hat))[0])                         # x + dt * (d + d_new) / 2
        b = float((dt / (2.0 * next_sigma))[0])
        t1 = ops.axpby(x, denoised, 1.0 + a, -a)
        t2 = ops.axpby(euler_step, denoised2, b, -b)
        return ops.axpby(t1, t2, 1.0, 1.0)


class DPMPP2MSampler(BaseDiffusionSampler):
    """sampling.py:408-485: multistep DPM-Solver++(2M); one evaluation per step."""

    def get_variables(self, sigma, next_sigma, previous_sigma=None):
        t, t_next = [to_neg_log_sigma(s) for s in (sigma, next_sigma)]
        h = t_next - t
        if previous_sigma is not None:
            h_last = t - to_neg_log_sigma(previous_sigma)
            return h, h_last / h, t, t_next
        return h, None, t, t_next

    def get_mult(self, h, r, t, t_next, previous_sigma):
        mult1 = to_sigma(t_next) / to_sigma(t)
        mult2 = (-h).expm1()
        if previous_sigma is not None:
            return mult1, mult2, 1 + 1 / (2 * r), 1 / (2 * r)
        return mult1, mult2

    def sampler_step(self, old_denoised, previous_sigma, sigma, next_sigma, denoiser, x, cond, uc=None):
        denoised = self.denoise(x, denoiser, sigma, cond, uc)
        h, r, t, t_next = self.get_variables(sigma, next_sigma, previous_sigma)
        mult = [float(m[0]) for m in self.get_mult(h, r, t, t_next, previous_sigma)]
        if old_denoised is None or torch.sum(next_sigma) < 1e-14:
            return ops.axpby(x, denoised, mult[0], -mult[1]), denoised
        denoised_d = ops.axpby(denoised, old_denoised, mult[2], -mult[3])
        return ops.axpby(x, denoised_d, mult[0], -mult[1]), denoised        # next_sigma > 0 here: the advanced branch

    def __call__(self, denoiser, x, cond, uc=None, num_steps=None, **kwargs):
        x, s_in, sigmas, num_sigmas, cond, uc = self.prepare_sampling_loop(x, cond, uc, num_steps)
        old_denoised = None
        for i in self.get_sigma_gen(num_sigmas):
            x, old_denoised = self.sampler_step(old_denoised, None if i == 0 else s_in * sigmas[i - 1], s_in * sigmas[i],
                                                s_in * sigmas[i + 1], denoiser, x, cond, uc=uc)
        return x


def linear_multistep_coeff(order, t, i, j, epsrel=1e-4):
    """sampling_utils.py:12-24 (Adams-Bashforth weights by quadrature of the Lagrange basis)."""
    from scipy import integrate
    if order - 1 > i:
        raise ValueError(f"Order {order} too high for step {i}")

    def fn(tau):
        prod = 1.0
        for k in range(order):
            if j == k:
                continue
            prod *= (tau - t[i - k]) / (t[i - j] - t[i - k])
        return prod

    return integrate.quad(fn, t[i], t[i + 1], epsrel=epsrel)[0]


class LinearMultistepSampler(BaseDiffusionSampler):
    """sampling.py:268-304 (LMS, order 4 by default)."""

    def __init__(self, order=4, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.order = order

    def __call__(self, denoiser, x, cond, uc=None, num_steps=None, **kwargs):
        x, s_in, sigmas, num_sigmas, cond, uc = self.prepare_sampling_loop(x, cond, uc, num_steps)
        ds = []
        sigmas_cpu = sigmas.detach().cpu().numpy()
        for i in self.get_sigma_gen(num_sigmas):
            sigma = s_in * sigmas[i]
            denoised = self.denoise(x, denoiser, sigma, cond, uc)
            inv = 1.0 / float(sigma[0])
            ds.append(ops.axpby(x, denoised, inv, -inv))                # to_d: (x - denoised) / sigma
            if len(ds) > self.order:
                ds.pop(0)
            cur_order = min(i + 1, self.order)
            coeffs = [linear_multistep_coeff(cur_order, sigmas_cpu, i, j) for j in range(cur_order)]
            for coeff, d in zip(coeffs, reversed(ds)):
                x = ops.axpby(x, d, 1.0, float(coeff))
        return x
