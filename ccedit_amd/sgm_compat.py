"""Helpers that derive state-dict specs (key -> shape) from the build's own module tree, so synthetic
name-keyed weights (utils/synth.py) can be generated without the reference present."""
from __future__ import annotations

from typing import Dict, List, Tuple

import torch

from .network import ControlledUNetModel3DTV2V, OpenAIWrapperControlLDM3DTV2V
from .vae import AutoencoderKLInferenceWrapper


def network_params(model_channels=320, num_heads=8, context_dim=768, in_channels=4, out_channels=4,
                   attention_resolutions=(4, 2, 1), num_res_blocks=2, channel_mult=(1, 2, 4, 4), hint_channels=3,
                   control_scales=1.0, crossframe=False, **ignored) -> dict:
    """network_config.params of configs/inference_ccedit/keyframe_no2ndca_depthmidas.yaml:25-56."""
    common = dict(use_checkpoint=False, in_channels=in_channels, model_channels=model_channels,
                  attention_resolutions=list(attention_resolutions), num_res_blocks=num_res_blocks,
                  channel_mult=list(channel_mult), num_heads=num_heads, use_spatial_transformer=True,
                  transformer_depth=1, context_dim=context_dim, legacy=False)
    cn = dict(common, hint_channels=hint_channels, control_scales=control_scales)
    p = dict(common, out_channels=out_channels, disable_temporal_text_ca=True,
             controlnet_config=dict(target="sgm.modules.diffusionmodules.controlmodel.ControlNet2D", params=cn))
    if crossframe:       # TVI2V: keyframe_ref_cp_no2ndca_add_cfca_depthzoe.yaml:32-90
        p.update(enable_attention3d_crossframe=True, ST3DCA_ca_type="center_self",
                 controlnet_img_config=dict(target="sgm.modules.diffusionmodules.controlmodel.ControlNet2D",
                                            params=dict(cn, no_add_x=True, set_input_hint_block_as_identity=True,
                                                        disable_text_ca=True)))
    return p


def build_network(device="cpu", **cfg) -> OpenAIWrapperControlLDM3DTV2V:
    with torch.device(device):
        net = ControlledUNetModel3DTV2V(**network_params(**cfg))
    return OpenAIWrapperControlLDM3DTV2V(net)


def vae_params(ch=128, ch_mult=(1, 2, 4, 4), num_res_blocks=2, z_channels=4, out_ch=3, embed_dim=4, **ignored) -> dict:
    return dict(embed_dim=embed_dim, monitor="val/rec_loss", lossconfig=dict(target="torch.nn.Identity"),
                ddconfig=dict(double_z=True, z_channels=z_channels, resolution=256, in_channels=3, out_ch=out_ch, ch=ch,
                              ch_mult=list(ch_mult), num_res_blocks=num_res_blocks, attn_resolutions=[], dropout=0.0))


def build_vae(device="cpu", **cfg) -> AutoencoderKLInferenceWrapper:
    with torch.device(device):
        return AutoencoderKLInferenceWrapper(**vae_params(**cfg))


def build_network_spec(cfg: dict, prefix: str = "model.") -> List[Tuple[str, Tuple[int, ...]]]:
    w = build_network("meta", **cfg)
    return [(prefix + k, tuple(v.shape)) for k, v in w.state_dict().items()]


def build_vae_spec(cfg: dict, prefix: str = "first_stage_model.") -> List[Tuple[str, Tuple[int, ...]]]:
    v = build_vae("meta", **cfg)
    return [(prefix + k, tuple(t.shape)) for k, t in v.state_dict().items()]


def engine_config(crossframe: bool = False, vae_ch: int = 128, **net_cfg) -> dict:
    """The `model:` section of the reference's inference yamls as a dict (keyframe_no2ndca_depthmidas.yaml, or with
    crossframe=True keyframe_ref_cp_no2ndca_add_cfca_depthzoe.yaml), optionally at reduced width for tests."""
    dd = "sgm.modules.diffusionmodules."
    emb = [dict(is_trainable=False, input_key="txt", ucg_rate=0.5, target="sgm.modules.encoders.modules.FrozenCLIPEmbedder"),
           dict(is_trainable=False, input_key="control_hint",
                target="sgm.modules.encoders.modules." + ("DepthZoeEncoder" if crossframe else "DepthMidasEncoder"))]
    if crossframe:
        emb.append(dict(is_trainable=False, input_key="cond_img", ucg_rate=0.0, target="sgm.modules.encoders.modules.VAEEmbedder"))
    return dict(target="sgm.models.diffusion.VideoDiffusionEngineTV2V", params=dict(
        use_ema=False, scale_factor=0.18215, disable_first_stage_autocast=True, log_keys=["txt"], freeze_model="spatial",
        denoiser_config=dict(target=dd + "denoiser.DiscreteDenoiser", params=dict(
            num_idx=1000, weighting_config=dict(target=dd + "denoiser_weighting.EpsWeighting"),
            scaling_config=dict(target=dd + "denoiser_scaling.EpsScaling"),
            discretization_config=dict(target=dd + "discretizer.LegacyDDPMDiscretization"))),
        network_config=dict(target=dd + "controlmodel.ControlledUNetModel3DTV2V",
                            params=network_params(crossframe=crossframe, **net_cfg)),
        conditioner_config=dict(target="sgm.modules.GeneralConditioner", params=dict(emb_models=emb)),
        first_stage_config=dict(target="sgm.models.autoencoder.AutoencoderKLInferenceWrapper", params=vae_params(ch=vae_ch))))
