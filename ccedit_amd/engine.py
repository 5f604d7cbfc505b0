"""VideoDiffusionEngineTV2V: composition of network wrapper, denoiser, conditioner and first stage.

Reference: sgm/models/diffusion.py:47-163 (DiffusionEngine), 361-385 (VideoDiffusionEngine), 560-606
(VideoDiffusionEngineTV2V).  Only the inference composition is kept: `.model` (wrapper around the
network), `.denoiser`, `.conditioner`, `.first_stage_model`, `.scale_factor`, `decode_first_stage`,
`encode_first_stage`.  Training (optimizers, losses, EMA, Lightning hooks) is out of scope.
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn as nn

from .config import get_obj_from_str, instantiate_from_config
from .network import OpenAIWrapperControlLDM3DTV2V

OPENAIUNETWRAPPERCONTROLLDM3DTV2V = "sgm.modules.diffusionmodules.wrappers.OpenAIWrapperControlLDM3DTV2V"


class VideoDiffusionEngineTV2V(nn.Module):
    def __init__(self, network_config, denoiser_config, first_stage_config, conditioner_config=None,
                 sampler_config=None, network_wrapper: Optional[str] = None, ckpt_path=None, use_ema: bool = False,
                 scale_factor: float = 1.0, disable_first_stage_autocast: bool = False, input_key: str = "jpg",
                 log_keys=None, no_cond_log: bool = False, compile_model: bool = False, freeze_model=None, **ignored):
        super().__init__()
        if use_ema:
            raise NotImplementedError("EMA is a training feature (out of scope)")
        self.log_keys, self.input_key = log_keys, input_key
        model = instantiate_from_config(network_config)
        wrapper_cls = get_obj_from_str(network_wrapper) if network_wrapper else OpenAIWrapperControlLDM3DTV2V
        self.model = wrapper_cls(model, compile_model=compile_model)
        self.denoiser = instantiate_from_config(denoiser_config)
        self.sampler = instantiate_from_config(sampler_config) if sampler_config is not None else None
        self.conditioner = None
        if conditioner_config is not None:
            self.conditioner = instantiate_from_config(conditioner_config)
        self.first_stage_model = instantiate_from_config(first_stage_config).eval()
        self.scale_factor = scale_factor
        self.disable_first_stage_autocast = disable_first_stage_autocast
        # In the reference this flag IS the first stage's precision (diffusion.py:151-156: autocast off = fp32 tensors and products; the
        # shipped yamls set it).  Here the first stage runs on the bf16 kernels by default — the dtype BASELINE.json quotes the path in —
        # and policy vae_fp32 selects the fp32 kernels: 1 = always, 2 = exactly when the yaml's flag says so.
        from . import policy
        if policy.get("vae_fp32") == 2 and hasattr(self.first_stage_model, "precision"):
            self.first_stage_model.precision = "fp32" if disable_first_stage_autocast else "bf16"
        self.setup_vaeembedder()
        self.use_ema = False
        if ckpt_path is not None:
            self.init_from_ckpt(ckpt_path)

    def setup_vaeembedder(self) -> None:
        """diffusion.py:375-385: the TVI2V `VAEEmbedder` shares the engine's first stage (not a sub-module of it)."""
        for e in (self.conditioner.embedders if self.conditioner is not None else []):
            if e.__class__.__name__ == "VAEEmbedder":
                e.__dict__["first_stage_model"] = self.first_stage_model        # plain attribute: no second registration
                e.disable_first_stage_autocast = self.disable_first_stage_autocast
                e.scale_factor = self.scale_factor
                e.freeze()

    # -- weights -------------------------------------------------------------------------------
    def init_from_ckpt(self, path: str) -> None:
        """diffusion.py:113-137 / 582-606: strict=False load of a .ckpt pickle or .safetensors."""
        if path.endswith("ckpt"):
            sd = torch.load(path, map_location="cpu")["state_dict"]
        elif path.endswith("safetensors"):
            from safetensors.torch import load_file
            sd = load_file(path)
        else:
            raise NotImplementedError
        sd = {(k[len("_forward_module."):] if k.startswith("_forward_module.") else k): v for k, v in sd.items()}
        missing, unexpected = self.load_state_dict(sd, strict=False)
        print(f"Restored from {path} with {len(missing)} missing and {len(unexpected)} unexpected keys")

    def pack(self, device=None):
        """Build the kernels' bf16 weight layout on `device` (call after load_state_dict)."""
        device = torch.device("cuda") if device is None else torch.device(device)
        self.model.diffusion_model.pack(device)
        self.first_stage_model.pack(device)
        for e in (self.conditioner.embedders if self.conditioner is not None else []):
            if hasattr(e, "pack"):
                e.pack(device)                       # FrozenCLIPEmbedder: the text transformer's kernel operands
        self.denoiser.to(device)
        return self

    @property
    def device(self):
        return next(self.model.parameters()).device

    # -- first stage ---------------------------------------------------------------------------
    @torch.no_grad()
    def decode_first_stage(self, z):
        """diffusion.py:151-156: z / scale_factor -> first_stage_model.decode."""
        from . import ops
        zf = z.float().contiguous()
        zs = ops.axpby(zf, zf, 1.0 / self.scale_factor, 0.0)
        return self.first_stage_model.decode(zs)

    @torch.no_grad()
    def encode_first_stage(self, x, noise=None):
        """diffusion.py:158-163: scale_factor * first_stage_model.encode(x) (a posterior SAMPLE; `noise` overrides the
        reference's CPU-generator draw for reproducible tests)."""
        from . import ops
        z = self.first_stage_model.encode(x, noise=noise)
        return ops.axpby(z, z, float(self.scale_factor), 0.0)
