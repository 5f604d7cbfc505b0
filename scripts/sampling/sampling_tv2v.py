#!/usr/bin/env python3
"""TV2V sampling entry point on the MI355X path — counterpart of the reference's
scripts/sampling/sampling_tv2v.py (same flag names for the options on the path; core loop = its lines 333-470,
including the `--prior_coefficient_x` noise prior and the `--sdedit_denoise_strength` branch).

Conditioning producers (CLIP text encoder, MiDaS depth, video decoding) are outside this build (weights / codecs
unavailable offline): pass precomputed tensors with --cond_path — a .pt/.safetensors holding
`tokens`, `tokens_uc` (1,77) int64 CLIP token ids (or precomputed `crossattn`, `crossattn_uc` (1,77,768)), `control_hint` (1,3,T,H,W) in [-1,1] and, for the prior / SDEdit options,
`keyframes` (1,3,T,H,W) in [-1,1] — or use --synthetic for seeded random tensors of the right shapes.  They enter
through `model.conditioner.get_unconditional_conditioning` exactly like the reference's batch dicts.
Outputs: `<save_path>/result/sample_XXXX.npy` (frames in [0,1], (T,H,W,3)) + `log_info.json` resume-skip list.
"""
from __future__ import annotations

import argparse
import os
import sys
import time

import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GUIDER = "sgm.modules.diffusionmodules.guiders.VanillaCFGTV2V"


def add_common_args(p: argparse.ArgumentParser) -> None:
    p.add_argument("--seed", type=int, default=42)
    p.add_argument("--config_path", type=str, default="")
    p.add_argument("--ckpt_path", type=str, default="")
    p.add_argument("--basemodel_path", type=str, default="", help="load a new base model instead of original sd-1.5")
    p.add_argument("--lora_path", type=str, default="")
    p.add_argument("--lora_strength", type=float, default=0.8)
    p.add_argument("--vae_path", type=str, default="")
    p.add_argument("--cond_path", type=str, default="", help="precomputed conditioning tensors (see module docstring)")
    p.add_argument("--synthetic", action="store_true", help="seeded random conditioning + name-keyed synthetic weights")
    p.add_argument("--video_path", type=str, default="", help="directory of frame images or .gif: source of `keyframes`")
    p.add_argument("--original_fps", type=int, default=20)
    p.add_argument("--target_fps", type=int, default=3)
    p.add_argument("--save_type", type=str, default="npy", choices=["npy", "gif"])
    p.add_argument("--save_path", type=str, default="outputs/demo/tv2v")
    p.add_argument("--H", type=int, default=256)
    p.add_argument("--W", type=int, default=384)
    p.add_argument("--num_keyframes", type=int, default=9)
    p.add_argument("--prompt", type=str, default="")
    p.add_argument("--negative_prompt", type=str, default="ugly, low quality")
    p.add_argument("--add_prompt", type=str, default="masterpiece, high quality")
    p.add_argument("--tokenizer_path", type=str, default="",
                   help="directory with the CLIP tokenizer files (vocab.json, merges.txt) of openai/clip-vit-large-patch14: with it "
                        "--prompt / --negative_prompt / --add_prompt are tokenised here as in the reference script; without it the "
                        "conditioning tensors must carry token ids or embeddings (--cond_path)")
    p.add_argument("--sample_steps", type=int, default=50)
    p.add_argument("--sampler_name", type=str, default="EulerEDMSampler")       # the reference script's default
    p.add_argument("--discretization_name", type=str, default="LegacyDDPMDiscretization")
    p.add_argument("--cfg_scale", type=float, default=7.5)
    p.add_argument("--prior_coefficient_x", type=float, default=0.0)
    p.add_argument("--prior_coefficient_noise", type=float, default=1.0)
    p.add_argument("--sdedit_denoise_strength", type=float, default=0.0)
    p.add_argument("--inpainting_mode", action="store_true", help="inpainting mode")
    p.add_argument("--num_samples", type=int, default=1)
    p.add_argument("--noise_seed", type=int, default=None,
                   help="(not in the reference script) draw the samplers' per-step noise from a CPU generator with this seed instead of "
                        "torch.randn_like on the device: the same clip on any GPU / against the CPU oracle (tests)")
    p.add_argument("--disable_check_repeat", action="store_true")


def build_model(args):
    from ccedit_amd.utils.synth import fill_module_
    from scripts.sampling.util import create_model
    if not args.config_path:
        raise SystemExit("--config_path is required (e.g. configs/inference_ccedit/keyframe_no2ndca_depthmidas.yaml)")
    dev = torch.device("cuda")
    from scripts.sampling.util import load_lora_file, load_vae_file, model_load_ckpt
    if getattr(args, "tokenizer_path", ""):
        os.environ["CCEDIT_CLIP_TOKENIZER"] = args.tokenizer_path      # read by FrozenCLIPEmbedder when the yaml builds it
    model = create_model(args.config_path, dev)
    if args.ckpt_path:
        model_load_ckpt(model, path=args.ckpt_path)
    elif args.synthetic:
        fill_module_(model.model, prefix="model.")
        fill_module_(model.first_stage_model, prefix="first_stage_model.")
        fill_module_(model.conditioner, prefix="conditioner.")
    else:
        raise SystemExit("need --ckpt_path or --synthetic")
    # sampling_tv2v.py:190-262: optional base-model swap, LoRA merge at --lora_strength, replacement VAE — all weight-space,
    # before the kernels' operands are packed
    if args.basemodel_path:
        model_load_ckpt(model, args.basemodel_path, True)
    if args.lora_path:
        load_lora_file(model, args.lora_path, args.lora_strength)
    if args.vae_path:
        load_vae_file(model, args.vae_path)
    model.pack(dev)
    args.context_dim = model.model.diffusion_model.context_dim     # 768 for the shipped configs
    return model, dev


def conditioning_tensors(args, g: torch.Generator, need_frames: bool, need_ref: bool = False):
    from scripts.sampling.util import load_conditioning
    T = args.num_keyframes
    if args.cond_path:
        cond = load_conditioning(args.cond_path)
    else:
        if args.context_dim == 768:      # shipped configs: seeded random token ids through the CLIP text encoder
            text = dict(tokens=torch.randint(0, 49406, (1, 77), generator=g), tokens_uc=torch.randint(0, 49406, (1, 77), generator=g))
        else:                            # reduced test configs: a precomputed embedding of the network's context width
            text = dict(crossattn=torch.randn(1, 77, args.context_dim, generator=g),
                        crossattn_uc=torch.randn(1, 77, args.context_dim, generator=g))
        cond = dict(**text,
                    control_hint=(torch.rand(1, 1, T, args.H, args.W, generator=g) * 2 - 1).repeat(1, 3, 1, 1, 1))
        if need_frames:
            cond["keyframes"] = torch.rand(1, 3, T, args.H, args.W, generator=g) * 2 - 1
        if need_ref:
            cond["cond_img"] = torch.rand(1, 3, args.H, args.W, generator=g) * 2 - 1
    if args.video_path:           # sampling_tv2v.py:314-331: keyframes (T,3,H,W) -> (1,3,T,H,W)
        from scripts.sampling.util import load_video_keyframes
        kf = load_video_keyframes(args.video_path, args.original_fps, args.target_fps, T, (args.H, args.W))
        cond["keyframes"] = kf.permute(1, 0, 2, 3)[None].contiguous()
    if need_ref and getattr(args, "reference_path", ""):
        from scripts.sampling.util import load_img
        cond["cond_img"] = load_img(args.reference_path, (args.H, args.W))
    for k in (["keyframes"] if need_frames else []) + (["cond_img"] if need_ref else []):
        if k not in cond:
            raise SystemExit(f"--cond_path must hold `{k}` for the requested options")
    return cond


def text_inputs(cond, dev, args=None):
    """batch['txt'] for prompt / negative prompt: the strings themselves when --tokenizer_path is given (composed as
    sampling_tv2v.py:334-347: add_prompt + ", " + prompt; the negative prompt alone), else token ids (`tokens`, `tokens_uc`:
    (1,77) int64 -> CLIP text encoder on the GPU) or precomputed embeddings (`crossattn`, `crossattn_uc`)."""
    if args is not None and getattr(args, "tokenizer_path", ""):
        prompt = args.add_prompt + ", " + args.prompt if args.add_prompt else args.prompt
        return [prompt], [args.negative_prompt]
    if "tokens" in cond:
        return cond["tokens"].to(dev), cond["tokens_uc"].to(dev)
    return cond["crossattn"].to(dev), cond["crossattn_uc"].to(dev)


def save_result(args, tag, x):
    """sampling_tv2v.py:473-515: clamp to [0,1]; .npy frames (default) or an animated gif + frame grid."""
    from scripts.sampling.util import perform_save_locally_video, save_frames
    if args.save_type == "gif":
        perform_save_locally_video(os.path.join(args.save_path, "result"), torch.clamp((x + 1.0) / 2.0, 0.0, 1.0),
                                   fps=args.target_fps, savetype="gif")
    save_frames(args.save_path, tag, x)


def _cpu_noise(sampler, args) -> None:
    """--noise_seed: the ancestral / churn noise (`noise_sampler`, torch.randn_like(x) in the reference: sampling.py:100, 184) from a seeded
    CPU generator, one draw per call in the samplers' own order."""
    seed = getattr(args, "noise_seed", None)
    if seed is not None and hasattr(sampler, "noise_sampler"):
        gen = torch.Generator().manual_seed(int(seed))
        sampler.noise_sampler = lambda x: torch.randn(x.shape, generator=gen).to(x.device)


def sample_one(args, model, dev, c, uc, randn, keyframes=None, ref=None, prior_type="video"):
    """sampling_tv2v.py:361-470 for one clip."""
    from scripts.sampling.util import init_sampling, prior_latent, sdedit_start

    def denoiser(inp, sigma, cc):
        return model.denoiser(model.model, inp, sigma, cc)

    if getattr(args, "inpainting_mode", False):
        # the reference script raises here too (sampling_tv2v.py:385-386, 444-445: the mask is not a user input yet);
        # the loop itself is available as sampler.sample_inpainting(denoiser, x, c, x0=z, mask=mask, uc=uc)
        raise NotImplementedError
    if args.sdedit_denoise_strength == 0.0:
        if args.prior_coefficient_x != 0.0:
            randn = prior_latent(model, randn, args.prior_coefficient_x, args.prior_coefficient_noise, keyframes, ref, prior_type)
        sampler = init_sampling(sample_steps=args.sample_steps, sampler_name=args.sampler_name,
                                discretization_name=args.discretization_name, guider_config_target=GUIDER,
                                cfg_scale=args.cfg_scale)
        _cpu_noise(sampler, args)
        samples = sampler(denoiser, randn, c, uc=uc)
    else:
        assert 0.0 < args.sdedit_denoise_strength <= 1.0, "sdedit_denoise_strength should be in (0, 1]"
        assert args.prior_coefficient_x == 0, "prior_coefficient_x should be 0 when using sdedit_denoise_strength"
        sampler = init_sampling(sample_steps=args.sample_steps, sampler_name=args.sampler_name,
                                discretization_name=args.discretization_name, guider_config_target=GUIDER,
                                cfg_scale=args.cfg_scale, img2img_strength=args.sdedit_denoise_strength)
        _cpu_noise(sampler, args)
        samples = sampler(denoiser, sdedit_start(model, sampler, keyframes), cond=c, uc=uc)
    return model.decode_first_stage(samples)


def main():
    p = argparse.ArgumentParser()
    add_common_args(p)
    args = p.parse_args()
    torch.manual_seed(args.seed)
    torch.set_grad_enabled(False)
    from scripts.sampling.util import ResumeLog, save_frames
    model, dev = build_model(args)
    T, h, w = args.num_keyframes, args.H // 8, args.W // 8
    g = torch.Generator().manual_seed(args.seed)
    need_frames = args.prior_coefficient_x != 0.0 or args.sdedit_denoise_strength != 0.0
    cond = conditioning_tensors(args, g, need_frames)
    hint = cond["control_hint"].to(dev)
    txt, txt_uc = text_inputs(cond, dev, args)
    batch = {"txt": txt, "control_hint": hint}
    batch_uc = {"txt": txt_uc, "control_hint": hint.clone()}                              # uc keeps the SAME hint (:339-344)
    c, uc = model.conditioner.get_unconditional_conditioning(batch, batch_uc=batch_uc)
    keyframes = cond["keyframes"].to(dev) if need_frames else None
    log = ResumeLog(args.save_path)
    for i in range(args.num_samples):
        tag = f"sample_{i:04d}"
        if log.done(tag) and not args.disable_check_repeat:
            continue
        randn = torch.randn(1, 4, T, h, w, generator=g).to(dev)                           # CPU generator, like :363
        t0 = time.time()
        x = sample_one(args, model, dev, c, uc, randn, keyframes=keyframes)
        torch.cuda.synchronize()
        save_result(args, tag, x)
        print(f"{tag}: {T} frames {args.H}x{args.W} in {time.time() - t0:.2f}s")
        log.add(tag)


if __name__ == "__main__":
    main()
