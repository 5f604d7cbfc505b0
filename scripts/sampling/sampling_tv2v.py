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

Job mode (round 6) — the reference script's own surface (sampling_tv2v.py:45-72, 106-204, 264-515): `--prompt` + `--video_path`,
`--prompt_listpath` + `--video_listpath`, `--videos_directory` (sub-directory name = prompt) or the BalanceCC layout `--json_path` +
`--videos_root` (item["Video Type"] / item["Video Name"] + ".mp4", one job per item["Editing"][i]["Target Prompt"], saved under
<save_path>/<Video Type>/<Video Name>/<Target Prompt>); `--num_samples` repeats, `--batch_size` chunks (one CFG-doubled batch of
2 x bs clips per chunk), `--basemodel_listpath` / `--use_default` loop over base models, results in
<save_path>/<basemodel>/{original,result,control_hint}/ + log_info.json (video_paths, keyframes_paths; processed videos are skipped
unless --disable_check_repeat).  What the reference computes with networks that are not part of this build enters as data: the depth
of a clip from `<video>.depth.pt` / `--depth_root/<name>.pt` (raw depth (N,h,w) of ALL frames; normalised by the conditioner's
MiDaS / Zoe recipe), prompts through `--tokenizer_path`; `--synthetic` substitutes the frames' luminance and prompt-seeded token ids.
Launched under torch.distributed (RANK / WORLD_SIZE), the chunks are dealt round-robin to the ranks (BASELINE.json config 5:
independent clips, one per GPU, no collective — ccedit_amd.parallel.shard_clips).
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
    p.add_argument("--use_default", action="store_true", help="use default ckpt at first")
    p.add_argument("--basemodel_path", type=str, default="", help="load a new base model instead of original sd-1.5")
    p.add_argument("--basemodel_listpath", type=str, default="")
    p.add_argument("--lora_path", type=str, default="")
    p.add_argument("--lora_strength", type=float, default=0.8)
    p.add_argument("--vae_path", type=str, default="")
    p.add_argument("--cond_path", type=str, default="", help="precomputed conditioning tensors (see module docstring)")
    p.add_argument("--synthetic", action="store_true", help="seeded random conditioning + name-keyed synthetic weights")
    p.add_argument("--video_path", type=str, default="", help="directory of frame images or .gif: source of `keyframes`")
    p.add_argument("--prompt_listpath", type=str, default="")
    p.add_argument("--video_listpath", type=str, default="")
    p.add_argument("--videos_directory", type=str, default="", help="directory containing videos to be processed")
    p.add_argument("--json_path", type=str, default="", help="path to json file containing video paths and captions")
    p.add_argument("--videos_root", type=str, default="", help="path to the root of videos")
    p.add_argument("--depth_root", type=str, default="",
                   help="(not in the reference script, whose conditioner runs MiDaS / ZoeDepth itself) directory of <video name>.pt raw "
                        "depth tensors (N, h, w) over all frames of a video; default: <video_path>.depth.pt next to the video")
    p.add_argument("--detect_ratio", type=float, default=1.0)
    p.add_argument("--batch_size", type=int, default=4)
    p.add_argument("--original_fps", type=int, default=20)
    p.add_argument("--target_fps", type=int, default=3)
    p.add_argument("--save_type", type=str, default="npy", choices=["npy", "gif", "mp4"],
                   help="reference: gif | mp4 (default mp4).  mp4 needs a codec library (none offline: raises, as decoding does); npy = raw frames")
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


# ------------------------------------------------------------------------------------------
# Job mode: the reference script's list / directory / BalanceCC-json surface (sampling_tv2v.py:106-204, 264-515)
# ------------------------------------------------------------------------------------------
def expand_jobs(args, with_ref: bool = False):
    """The (prompt, video) jobs of one invocation, as the reference expands them — sampling_tv2v.py:106-160, and with `with_ref`
    sampling_tv2v_ref.py:124-194 (one reference image per job: --reference_path, or --reference_root/output-<Target Prompt>.png in the
    json layout, where jobs whose save directory exists are dropped).  Returns (prompts, video_paths, video_save_paths, ref_paths);
    video_save_paths is empty except in the json layout.  Same assertions and messages as the reference."""
    import json
    prompts, video_paths, video_save_paths, ref_paths = [], [], [], []
    assert not (args.prompt_listpath and args.videos_directory), (
        "Only one of prompt_listpath and videos_directory can be provided, "
        "but got prompt_listpath: {}, videos_directory: {}".format(args.prompt_listpath, args.videos_directory))
    if args.prompt_listpath:
        with open(args.prompt_listpath, "r") as f:
            prompts = [q.strip() for q in f.readlines()]
        assert args.video_listpath, (
            "video_listpath must be provided when prompt_listpath is provided, "
            "but got video_listpath: {}".format(args.video_listpath))
        with open(args.video_listpath, "r") as f:
            video_paths = [q.strip() for q in f.readlines()]
    elif args.videos_directory:
        for video_name in sorted(os.listdir(args.videos_directory)):       # (sorted: os.listdir order is arbitrary in the reference)
            video_path = os.path.join(args.videos_directory, video_name)
            if os.path.isdir(video_path):
                prompts.append(video_name)
                video_paths.append(video_path)
    elif args.json_path:
        assert args.videos_root != "", "videos_root must be provided when json_path is provided"
        if with_ref:
            assert getattr(args, "reference_root", "") != "", "reference_root must be provided when json_path is provided"
        with open(args.json_path, "r") as f:
            json_dict = json.load(f)
        for item in json_dict:
            video_path = os.path.join(args.videos_root, item["Video Type"], item["Video Name"] + ".mp4")
            for edit in item["Editing"]:
                video_save_path = os.path.join(args.save_path, item["Video Type"], item["Video Name"], edit["Target Prompt"])
                if with_ref and os.path.exists(video_save_path):           # sampling_tv2v_ref.py:165-167
                    print(f"video {video_save_path} exists, skip it.")
                    continue
                video_paths.append(video_path)
                prompts.append(edit["Target Prompt"])
                video_save_paths.append(video_save_path)
                if with_ref:
                    ref_paths.append(os.path.join(args.reference_root, "output-{}.png".format(edit["Target Prompt"])))
    else:
        assert args.prompt and args.video_path, (
            "prompt and video_path must be provided when prompt_listpath and videos_directory are not provided, "
            "but got prompt: {}, video_path: {}".format(args.prompt, args.video_path))
        prompts, video_paths = [args.prompt], [args.video_path]
    assert len(prompts) == len(video_paths), (
        "The number of prompts and video_paths must be the same, and you provided {} prompts and {} video_paths".format(
            len(prompts), len(video_paths)))
    if with_ref and not args.json_path:
        # sampling_tv2v_ref.py:192-193 keeps ONE reference path whatever the number of jobs (zip then drops all chunks but the first);
        # here the single --reference_path serves every job of the list
        ref_paths = [getattr(args, "reference_path", "")] * len(prompts)
    return prompts, video_paths, video_save_paths, ref_paths


def basemodel_list(args):
    """sampling_tv2v.py:186-204: the base models to loop over ("default" = the checkpoint as loaded)."""
    assert not (args.basemodel_path and args.basemodel_listpath), (
        "Only one of basemodel_path and basemodel_listpath can be provided, "
        "but got basemodel_path: {}, basemodel_listpath: {}".format(args.basemodel_path, args.basemodel_listpath))
    paths = []
    if args.basemodel_listpath:
        with open(args.basemodel_listpath, "r") as f:
            paths = [q.strip() for q in f.readlines()]
    if args.basemodel_path:
        paths = [args.basemodel_path]
    if args.use_default:
        paths = ["default"] + paths
    return paths or ["default"]


def resolve_video(video_path: str) -> str:
    """The json layout names <Video Name>.mp4; without a codec library (none offline) the same clip as a directory of frames
    <Video Name>/ or as <Video Name>.gif is taken instead.  Nothing else is rewritten."""
    if video_path.endswith(".mp4"):
        stem = video_path[:-4]
        if os.path.isdir(stem):
            return stem
        if os.path.exists(stem + ".gif"):
            return stem + ".gif"
    return video_path


def depth_frames(args, video_path: str, keyframes: torch.Tensor) -> torch.Tensor:
    """Raw depth (1, 1, T, H, W) of the clip's keyframes.  The reference's conditioner runs MiDaS dpt_hybrid / ZoeDepth on the RGB
    keyframes (encoders/modules.py:1289-1392); those networks are not part of this build, their output enters as data:
    --depth_root/<video name>.pt or <video_path>.depth.pt holding the depth of ALL frames (N, h, w) — the keyframe index rule and a
    bicubic resize are applied here like to the frames — or, with --synthetic, the keyframes' luminance."""
    from scripts.sampling.util import keyframe_indices
    stem = video_path[:-4] if video_path.endswith((".mp4", ".gif")) else video_path.rstrip("/")
    cands = ([os.path.join(args.depth_root, os.path.basename(stem) + ".pt")] if args.depth_root else []) + [stem + ".depth.pt"]
    T, H, W = keyframes.shape[2], keyframes.shape[3], keyframes.shape[4]
    for cpath in cands:
        if os.path.exists(cpath):
            d = torch.load(cpath, map_location="cpu").float()
            if d.dim() != 3:
                raise ValueError(f"{cpath}: expected raw depth (N, h, w) over all frames, got {tuple(d.shape)}")
            d = d[keyframe_indices(d.shape[0], args.original_fps, args.target_fps, T)]
            d = torch.nn.functional.interpolate(d[:, None], size=(H, W), mode="bicubic", align_corners=False)[:, 0]
            return d[None, None]
    if args.synthetic:
        kf = keyframes.float().cpu()
        return (0.299 * kf[:, 0:1] + 0.587 * kf[:, 1:2] + 0.114 * kf[:, 2:3])
    raise NotImplementedError(
        f"no depth for {video_path}: the depth annotators (MiDaS dpt_hybrid / ZoeDepth, encoders/modules.py:1289-1392) are not part of "
        f"this build — provide {cands[-1]} (raw depth (N, h, w) of all frames) or --depth_root")


def job_text(args, prompts, dev, context_dim: int):
    """batch['txt'] / batch_uc['txt'] of a chunk: strings when --tokenizer_path is given (add_prompt + ", " + prompt, the negative prompt
    bs times: sampling_tv2v.py:339-347); with --synthetic and no tokenizer files, token ids (embeddings at reduced context widths)
    seeded by the text — the same prompt always maps to the same ids."""
    import zlib
    full = [args.add_prompt + ", " + q if args.add_prompt else q for q in prompts]
    neg = [args.negative_prompt for _ in prompts]
    if args.tokenizer_path:
        return full, neg
    if not args.synthetic:
        raise SystemExit("prompts need --tokenizer_path (vocab.json / merges.txt of openai/clip-vit-large-patch14), or --synthetic")

    def one(text):
        g = torch.Generator().manual_seed(zlib.crc32(text.encode("utf-8")) & 0x7FFFFFFF)
        if context_dim == 768:
            return torch.randint(0, 49406, (1, 77), generator=g)
        return torch.randn(1, 77, context_dim, generator=g)
    return torch.cat([one(t) for t in full]).to(dev), torch.cat([one(t) for t in neg]).to(dev)


def _dist_rank_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def run_jobs(args, with_ref: bool = False) -> None:
    """sampling_tv2v.py:106-515 (sampling_tv2v_ref.py with `with_ref`): jobs -> num_samples -> chunks of batch_size -> for every base model,
    for every chunk: keyframes, conditioning, one CFG-doubled batch through the sampler, decode, original / result / control_hint +
    log_info.json."""
    import json
    from ccedit_amd.parallel import shard_clips
    from scripts.sampling.util import chunk, load_img, load_video_keyframes, model_load_ckpt, perform_save_locally_video
    prompts, video_paths, video_save_paths, ref_paths = expand_jobs(args, with_ref)
    num_samples, batch_size = args.num_samples, args.batch_size
    print("\nNumber of prompts: {}".format(len(prompts)))
    print("Generate {} samples for each prompt".format(num_samples))
    rep = lambda lst: [item for item in lst for _ in range(num_samples)]
    prompts_chunk = list(chunk(rep(prompts), batch_size))
    video_paths_chunk = list(chunk(rep(video_paths), batch_size))
    ref_paths_chunk = list(chunk(rep(ref_paths), batch_size)) if with_ref else [()] * len(prompts_chunk)
    # the json layout saves chunk idx under video_save_paths[idx] (sampling_tv2v.py:483): one job per chunk there
    if video_save_paths:
        assert batch_size == 1 and num_samples == 1, "the json layout saves one job per directory: use --batch_size 1 --num_samples 1"
    rank, world = _dist_rank_world()
    mine = set(shard_clips(len(prompts_chunk), rank, world))              # config 5: independent clips, one stream of chunks per GPU
    basemodels = basemodel_list(args)

    args_nobase = argparse.Namespace(**vars(args))
    args_nobase.basemodel_path = ""
    if world > 1:                                                         # one process per GPU (torch.distributed.run sets LOCAL_RANK)
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", rank)) % max(torch.cuda.device_count(), 1))
    model, dev = build_model(args_nobase)
    args.context_dim = args_nobase.context_dim
    T, H, W = args.num_keyframes, args.H, args.W
    g = torch.Generator().manual_seed(args.seed + 7919 * rank)
    for basemodel_idx, basemodel_path in enumerate(basemodels):
        print("-> base model idx: ", basemodel_idx)
        print("-> base model path: ", basemodel_path)
        if basemodel_path != "default" and basemodel_path:
            print("--> load a new base model from {}".format(basemodel_path))
            model_load_ckpt(model, basemodel_path, True)
            model.pack(dev)
        base_tag = basemodel_path.split("/")[-1].split(".")[0]
        log_name = "log_info.json" if world == 1 else f"log_info.rank{rank}.json"
        log_path = os.path.join(args.save_path, base_tag, log_name)
        if os.path.exists(log_path):
            with open(log_path, "r") as f:
                log_info = json.load(f)
        else:
            log_info = {"basemodel_path": basemodel_path, "lora_path": args.lora_path, "vae_path": args.vae_path,
                        "video_paths": [], "keyframes_paths": []}
        for idx, (cprompts, cvideos, crefs) in enumerate(zip(prompts_chunk, video_paths_chunk, ref_paths_chunk)):
            if idx not in mine:
                continue
            cprompts, cvideos, crefs = list(cprompts), list(cvideos), list(crefs)
            if not args.disable_check_repeat:                            # sampling_tv2v.py:297-309: leading videos already done are dropped
                while cvideos and cvideos[0] in log_info["video_paths"]:
                    print(f"video [{cvideos[0]}] has been processed, skip it.")
                    cprompts.pop(0)
                    cvideos.pop(0)
                    if crefs:
                        crefs.pop(0)
                if not cvideos:
                    continue
            bs = min(len(cprompts), batch_size)
            print(f"\nProgress: {idx} / {len(prompts_chunk)}. ")
            try:
                kfs = [load_video_keyframes(resolve_video(v), args.original_fps, args.target_fps, T, (H, W)).permute(1, 0, 2, 3)[None]
                       for v in cvideos]
            except Exception as e:                                       # (:312-330: a clip that does not load is reported and skipped)
                print(f"Error when loading video from  {cvideos}: {type(e).__name__}: {e}")
                continue
            keyframes = torch.cat(kfs, dim=0).to(dev)                    # (bs, 3, T, H, W) in [-1, 1]
            depth = torch.cat([depth_frames(args, v, k) for v, k in zip(cvideos, kfs)], dim=0).to(dev)
            txt, txt_uc = job_text(args, cprompts, dev, args.context_dim)
            batch = {"txt": txt, "control_hint": depth}
            batch_uc = {"txt": txt_uc, "control_hint": depth.clone()}     # uc keeps the SAME hint (:339-344)
            ref = None
            if with_ref:
                if getattr(args, "auto_ref_editing", False):
                    print("Conduct auto ref editing, args.reference_path is ignored.")
                    raise NotImplementedError                            # as the reference: sampling_tv2v_ref.py:366-369
                ref = torch.cat([load_img(r, (H, W)) for r in crefs], dim=0).to(dev)
                batch["cond_img"], batch_uc["cond_img"] = ref, ref.clone()
            c, uc = model.conditioner.get_unconditional_conditioning(batch, batch_uc=batch_uc)
            for k in c:
                if isinstance(c[k], torch.Tensor):
                    c[k], uc[k] = c[k][:bs].to(dev), uc[k][:bs].to(dev)
            randn = torch.randn(bs, 4, T, H // 8, W // 8, generator=g).to(dev)
            t0 = time.time()
            samples = sample_one(args, model, dev, c, uc, randn, keyframes=keyframes, ref=ref,
                                 prior_type=getattr(args, "prior_type", "video"))
            torch.cuda.synchronize()
            print(f"chunk {idx}: {bs} clip(s) of {T} frames {H}x{W} in {time.time() - t0:.2f}s")
            to01 = lambda v: (torch.clamp(v.float(), -1.0, 1.0) + 1.0) / 2.0
            save_path = video_save_paths[idx] if video_save_paths else os.path.join(args.save_path, base_tag)
            perform_save_locally_video(os.path.join(save_path, "original"), to01(keyframes), args.target_fps, args.save_type, save_grid=False)
            keyframes_paths = perform_save_locally_video(os.path.join(save_path, "result"), to01(samples), args.target_fps, args.save_type,
                                                         return_savepaths=True, save_grid=False)
            perform_save_locally_video(os.path.join(save_path, "control_hint"), to01(c["control_hint"]), args.target_fps, args.save_type,
                                       save_grid=False)
            print("Saved samples to {}. Enjoy.".format(save_path))
            log_info["video_paths"] += cvideos
            log_info["keyframes_paths"] += keyframes_paths
            os.makedirs(os.path.dirname(log_path), exist_ok=True)
            for out in {log_path, os.path.join(save_path, log_name)}:    # (the reference writes it under save_path; the loop reads it per base model)
                os.makedirs(os.path.dirname(out), exist_ok=True)
                with open(out, "w") as f:
                    json.dump(log_info, f, indent=4)
        if basemodel_idx + 1 < len(basemodels) and basemodel_path != "default":
            print("--> back to the original model: {}".format(args.ckpt_path))
            model, dev = build_model(args_nobase)                        # (:517-520: the checkpoint is loaded again before the next base model)


def job_mode(args) -> bool:
    return bool(args.prompt_listpath or args.videos_directory or args.json_path or (args.prompt and args.video_path))


def main():
    p = argparse.ArgumentParser()
    add_common_args(p)
    args = p.parse_args()
    torch.manual_seed(args.seed)
    torch.set_grad_enabled(False)
    if job_mode(args):
        return run_jobs(args)
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
