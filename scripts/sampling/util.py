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
SAMPLERS_BUILT = ("EulerEDMSampler", "HeunEDMSampler", "EulerAncestralSampler", "DPMPP2SAncestralSampler", "DPMPP2MSampler",
                  "LinearMultistepSampler")


def create_model(config_path: str, device="cuda"):
    config = load_config(config_path)
    with torch.device(device):
        return instantiate_from_config(config.model)


def get_discretization(discretization: str) -> dict:
    if discretization == "LegacyDDPMDiscretization":
        return {"target": _DD + "discretizer.LegacyDDPMDiscretization"}
    if discretization == "EDMDiscretization":
        return {"target": _DD + "discretizer.EDMDiscretization", "params": {"sigma_min": 0.03, "sigma_max": 14.61, "rho": 3.0}}
    raise ValueError(f"unknown discretization {discretization}")


def get_guider(guider_config_target=_DD + "guiders.VanillaCFG", scale=7.5) -> dict:
    return {"target": guider_config_target,
            "params": {"scale": scale, "dyn_thresh_config": {"target": _DD + "sampling_utils.NoDynamicThresholding"}}}


def get_sampler(sampler_name: str, steps: int, discretization_config: dict, guider_config: dict):
    if sampler_name not in SAMPLERS_BUILT:
        raise ValueError(f"unknown sampler {sampler_name}!")
    extra = {"EulerEDMSampler": dict(s_churn=0.0, s_tmin=0.0, s_tmax=999.0, s_noise=1.0),       # util.py:484-511
             "HeunEDMSampler": dict(s_churn=0.0, s_tmin=0.0, s_tmax=999.0, s_noise=1.0),
             "EulerAncestralSampler": dict(eta=1.0, s_noise=1.0),                              # :512-536
             "DPMPP2SAncestralSampler": dict(eta=1.0, s_noise=1.0),
             "DPMPP2MSampler": dict(),                                                         # :537-543
             "LinearMultistepSampler": dict(order=4)}[sampler_name]                            # :544-553
    return instantiate_from_config(dict(target=_DD + "sampling." + sampler_name, params=dict(
        num_steps=steps, discretization_config=discretization_config, guider_config=guider_config, verbose=True, **extra)))


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


def chunk(it, size):
    """util.py:355-357."""
    from itertools import islice
    it = iter(it)
    return iter(lambda: tuple(islice(it, size)), ())


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


# ------------------------------------------------------------------------------------------
# checkpoint ingestion: reference scripts/sampling/util.py:45-272
# ------------------------------------------------------------------------------------------
import re

# kohya "down_blocks_i / attentions_j" and "up_blocks_i / attentions_j" -> sgm input_blocks / output_blocks index
_LORA_IN = {(0, 0): 1, (0, 1): 2, (1, 0): 4, (1, 1): 5, (2, 0): 7, (2, 1): 8}
_LORA_OUT = {(1, 0): 3, (1, 1): 4, (1, 2): 5, (2, 0): 6, (2, 1): 7, (2, 2): 8, (3, 0): 9, (3, 1): 10, (3, 2): 11}
_RE_TE = re.compile(r"^lora_te_text_model_encoder_layers_(\d+)_(self_attn_([qkv]|out)_proj|mlp_(fc1|fc2))$")
_RE_UNET = re.compile(r"^lora_unet_(?:(down|up)_blocks_(\d+)|mid_block)_attentions_(\d+)_(.+)$")
_RE_TAIL = re.compile(r"^(?:proj_(in|out)|transformer_blocks_(\d+)_(?:(attn[12])_to_(?:([qkv])|out_(\d+))|ff_net_(\d+)(?:_(proj))?))$")


def lora_target_key(key: str) -> str:
    """State-dict key of the weight a kohya-format LoRA tensor (`<module>.lora_up/down.weight`) is merged into."""
    mod = key.split(".")[0]
    m = _RE_TE.match(mod)
    if m:
        layer = m.group(1)
        leaf = f"self_attn.{m.group(3)}_proj" if m.group(3) else f"mlp.{m.group(4)}"
        return f"conditioner.embedders.0.transformer.text_model.encoder.layers.{layer}.{leaf}.weight"
    m = _RE_UNET.match(mod)
    t = _RE_TAIL.match(m.group(4)) if m else None
    if not (m and t):
        raise ValueError("Unknown key: ", key)
    if m.group(1) is None:
        base = "model.diffusion_model.middle_block.1"
    else:
        table, name = (_LORA_IN, "input_blocks") if m.group(1) == "down" else (_LORA_OUT, "output_blocks")
        base = f"model.diffusion_model.{name}.{table[(int(m.group(2)), int(m.group(3)))]}.1"
    if t.group(1):
        return f"{base}.proj_{t.group(1)}.weight"
    tb = f"{base}.transformer_blocks.{t.group(2)}"
    if t.group(3):
        return f"{tb}.{t.group(3)}.to_{t.group(4)}.weight" if t.group(4) else f"{tb}.{t.group(3)}.to_out.{t.group(5)}.weight"
    return f"{tb}.ff.net.{t.group(6)}" + (".proj" if t.group(7) else "") + ".weight"


def convert_load_lora(sd_state_dict: Dict[str, torch.Tensor], state_dict: Dict[str, torch.Tensor],
                      LORA_PREFIX_UNET="lora_unet", LORA_PREFIX_TEXT_ENCODER="lora_te", alpha=0.6):
    """util.py:115-272: W += alpha * (lora_up @ lora_down) for every LoRA pair, in place in `sd_state_dict`
    (1x1-conv projections keep their (O, I, 1, 1) shape; `.alpha` entries are ignored like the reference does)."""
    if (LORA_PREFIX_UNET, LORA_PREFIX_TEXT_ENCODER) != ("lora_unet", "lora_te"):
        raise NotImplementedError("non-default LoRA prefixes")
    done = set()
    for key in state_dict:
        if ".alpha" in key or key in done:
            continue
        up_key = key.replace("lora_down", "lora_up")
        down_key = key.replace("lora_up", "lora_down")
        up, down = state_dict[up_key].to(torch.float32), state_dict[down_key].to(torch.float32)
        target = lora_target_key(key)
        if up.dim() == 4:
            delta = torch.mm(up.squeeze(3).squeeze(2), down.squeeze(3).squeeze(2)).unsqueeze(2).unsqueeze(3)
        else:
            delta = torch.mm(up, down)
        sd_state_dict[target] += alpha * delta
        done.update((up_key, down_key))
    return sd_state_dict


def read_checkpoint(path: str) -> Dict[str, torch.Tensor]:
    if path.endswith(("ckpt", ".pt", ".pth")):
        sd = torch.load(path, map_location="cpu")
        if "deepspeed" in path:
            return {k.replace("_forward_module.", ""): v for k, v in sd.items()}
        return sd["state_dict"] if "state_dict" in sd else sd
    if path.endswith("safetensors"):
        from safetensors.torch import load_file
        return load_file(path)
    raise NotImplementedError(f"Unknown checkpoint format: {path}")


def remap_checkpoint_keys(sd: Dict[str, torch.Tensor], newbasemodel: bool = False) -> Dict[str, torch.Tensor]:
    """util.py:63-82: VAE copies nested under conditioner embedders lose that prefix; with `newbasemodel` (a plain
    SD-1.5 checkpoint as base) `cond_stage_model.*` becomes `conditioner.embedders.0.*`."""
    out = {}
    for k, v in sd.items():
        if k.startswith("conditioner.embedders.") and "first_stage_model" in k:
            k = k[k.find("first_stage_model"):]
        if newbasemodel and "cond_stage_model" in k:
            k = k.replace("cond_stage_model", "conditioner.embedders.0")
        out[k] = v
    return out


def model_load_ckpt(model, path: str, newbasemodel: bool = False):
    """util.py:45-112: non-strict load with the key surgery above; LoRA tensors embedded in the checkpoint
    (`lora*` keys, e.g. majicmixRealistic) are merged at alpha = 0.8 like the reference does, then loaded."""
    sd = remap_checkpoint_keys(read_checkpoint(path), newbasemodel)
    lora = {k: v for k, v in sd.items() if k.startswith("lora")}
    if lora:
        for k in lora:
            del sd[k]
        convert_load_lora(sd_state_dict=sd, state_dict=lora, alpha=0.8)
    missing, unexpected = model.load_state_dict(sd, strict=False)
    if newbasemodel:
        unwanted = ["temporal", "controlnet", "conditioner.embedders.1."]
        missing = [k for k in missing if all(s not in k for s in unwanted)]
    print(f"Restored from {path} with {len(missing)} missing and {len(unexpected)} unexpected keys")
    if missing:
        print(f"Missing Keys: {missing}")
    if unexpected:
        print(f"Unexpected Keys: {unexpected}")
    return model


def load_lora_file(model, lora_path: str, strength: float) -> None:
    """sampling_tv2v.py:212-236: merge a kohya .safetensors LoRA into the model's weights at `strength`."""
    if not lora_path.endswith(".safetensors"):
        raise NotImplementedError
    from safetensors.torch import load_file
    lora_sd = load_file(lora_path)
    if not all("lora" in k for k in lora_sd):
        raise ValueError(f"The model you provided in [{lora_path}] is not a LoRA model. ")
    sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    model.load_state_dict(convert_load_lora(sd, lora_sd, alpha=strength))


def load_vae_file(model, vae_path: str) -> None:
    """sampling_tv2v.py:238-260: replacement first stage (.pt with `state_dict`, or .safetensors)."""
    if vae_path.endswith(".pt"):
        vae_sd = torch.load(vae_path, map_location="cpu")["state_dict"]
    elif vae_path.endswith(".safetensors"):
        from safetensors.torch import load_file
        vae_sd = load_file(vae_path)
    else:
        raise ValueError("Cannot load vae model from {}".format(vae_path))
    print("msg of loading vae: ", model.first_stage_model.load_state_dict(vae_sd, strict=False))


# ------------------------------------------------------------------------------------------
# frame / video I/O: reference scripts/sampling/util.py:288-382, 689-762
# Image files and GIFs go through Pillow; mp4 needs a codec library (decord / cv2 / imageio-ffmpeg) that is not
# installed here and raises.
# ------------------------------------------------------------------------------------------
def load_img(p_cond_img: str, size: tuple = None) -> torch.Tensor:
    """util.py:360-382: image file -> (1, 3, H, W) in [-1, 1], optional bicubic resize to size = (H, W)."""
    from PIL import Image
    img = Image.open(p_cond_img)
    if size:
        assert len(size) == 2, "size should be (H, W)"
        h, w = size
        img = img.resize((w, h), Image.BICUBIC)
    t = torch.from_numpy(np.array(img)).permute(2, 0, 1).unsqueeze(0).float() / 255.0
    return torch.clamp(t * 2.0 - 1.0, -1.0, 1.0)


def keyframe_indices(num_allframes: int, original_fps: int, target_fps: int, num_keyframes: int) -> np.ndarray:
    """util.py:708-720 / 735-745: every round(original_fps / target_fps)-th frame, the first `num_keyframes` of them;
    when the video is too short, `num_keyframes` indices spread evenly over it instead."""
    gap = int(np.round(original_fps / target_fps).astype(int))
    assert gap > 0, f"gap {gap} should be positive."
    idx = list(range(0, num_allframes, gap))
    if len(idx) < num_keyframes:
        print("[WARNING]: not enough keyframes, use linspace instead. "
              f"len(keyindexs): [{len(idx)}] < num_keyframes [{num_keyframes}]")
        return np.linspace(0, num_allframes - 1, num_keyframes).astype(int)
    return np.asarray(idx[:num_keyframes])


def HWC3(x: np.ndarray) -> np.ndarray:
    """util.py:559-576: uint8 image with 1 / 3 / 4 channels (or none) -> 3 channels (alpha blended on white)."""
    assert x.dtype == np.uint8
    if x.ndim == 2:
        x = x[:, :, None]
    assert x.ndim == 3
    c = x.shape[2]
    assert c in (1, 3, 4)
    if c == 3:
        return x
    if c == 1:
        return np.concatenate([x, x, x], axis=2)
    color = x[:, :, 0:3].astype(np.float32)
    alpha = x[:, :, 3:4].astype(np.float32) / 255.0
    return (color * alpha + 255.0 * (1.0 - alpha)).clip(0, 255).astype(np.uint8)


def load_video_keyframes(video_path: str, original_fps: int, target_fps: int, num_keyframes: int, size: tuple = None) -> torch.Tensor:
    """util.py:689-762: directory of frame images or a .gif -> keyframes (T, 3, H, W) in [-1, 1]."""
    if os.path.isdir(video_path):
        files = sorted(os.listdir(video_path))
        idx = keyframe_indices(len(files), original_fps, target_fps, num_keyframes)
        return torch.cat([load_img(os.path.join(video_path, files[i]), size) for i in idx], dim=0)
    if video_path.endswith(".gif"):
        from PIL import Image, ImageSequence
        frames = np.stack([HWC3(np.array(fr.convert("RGBA") if fr.mode == "P" and "transparency" in fr.info else fr.convert("RGB")))
                           for fr in ImageSequence.Iterator(Image.open(video_path))], axis=0)
        frames = torch.from_numpy(frames).permute(0, 3, 1, 2).float() / 255.0
        frames = frames[keyframe_indices(frames.shape[0], original_fps, target_fps, num_keyframes)]
        frames = torch.clamp(frames * 2.0 - 1.0, -1.0, 1.0)
        if size:
            assert len(size) == 2, "size should be (H, W)"
            frames = torch.nn.functional.interpolate(frames, size=size, mode="bicubic", align_corners=False)
        return frames
    if video_path.endswith(".mp4"):
        raise NotImplementedError("mp4 decoding needs decord / cv2 / imageio-ffmpeg, none of which is installed; "
                                  "extract the frames to a directory of images (or a .gif) instead")
    raise ValueError("Unsupported video format. Only support dirctory, .mp4 and .gif.")


def perform_save_locally_video(save_path: str, samples: torch.Tensor, fps: int, savetype: str = "gif",
                               return_savepaths: bool = False, save_grid: bool = True):
    """util.py:288-352: samples (B, 3, T, H, W) in [0, 1] -> <save_path>/gif/animation-XXXX.gif (+ grid/grid-XXXX.png:
    the T frames side by side).  savetype='mp4' needs a codec library and raises."""
    from PIL import Image
    assert samples.dim() == 5, "Expected samples to have shape (B, C, T, H, W)"
    assert savetype in ["gif", "mp4", "npy"]
    if savetype == "mp4":
        raise NotImplementedError("mp4 encoding needs imageio-ffmpeg / cv2, not installed here: use savetype='gif'")
    if savetype == "npy":          # (not in the reference) the frames themselves: <save_path>/npy/frames-XXXX.npy, (T, H, W, C) float32 in [0, 1]
        os.makedirs(os.path.join(save_path, "npy"), exist_ok=True)
        count = len(os.listdir(os.path.join(save_path, "npy")))
        savepaths = []
        for sample in samples:
            savepath = os.path.join(save_path, "npy", f"frames-{count:04}.npy")
            np.save(savepath, sample.detach().float().cpu().permute(1, 2, 3, 0).numpy())
            count += 1
            savepaths.append(savepath)
        return savepaths if return_savepaths else None
    os.makedirs(os.path.join(save_path, savetype), exist_ok=True)
    count = len(os.listdir(os.path.join(save_path, savetype)))
    if save_grid:
        os.makedirs(os.path.join(save_path, "grid"), exist_ok=True)
        count_grid = len(os.listdir(os.path.join(save_path, "grid")))
    savepaths = []
    for sample in samples:
        frames_f = sample.detach().float().cpu().permute(1, 2, 3, 0).numpy()              # (T, H, W, C)
        if save_grid:
            # torchvision.utils.save_image(normalize=False, padding=0): x * 255 + 0.5, clamp, uint8
            grid = np.concatenate(list(np.clip(frames_f * 255.0 + 0.5, 0, 255).astype(np.uint8)), axis=1)
            Image.fromarray(grid).save(os.path.join(save_path, "grid", f"grid-{count_grid:04}.png"))
            count_grid += 1
        frames = [Image.fromarray(f) for f in (255.0 * frames_f).astype(np.uint8)]
        savepath = os.path.join(save_path, "gif", f"animation-{count:04}.gif")
        frames[0].save(savepath, save_all=True, append_images=frames[1:], duration=int(round(1000.0 / fps)), loop=0)
        count += 1
        savepaths.append(savepath)
    return savepaths if return_savepaths else None
