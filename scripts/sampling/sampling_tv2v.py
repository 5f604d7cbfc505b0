#!/usr/bin/env python3
"""TV2V sampling entry point on the MI355X path — counterpart of the reference's
scripts/sampling/sampling_tv2v.py (same flag names for the hot-path options; core loop = its lines 333-470).

Conditioning producers (CLIP text encoder, MiDaS depth) are outside this build (weights unavailable offline,
SURVEY.md §2 row 14): pass precomputed tensors with --cond_path (a .pt/.safetensors holding `crossattn`,
`crossattn_uc` (1,77,768) and `control_hint` (1,3,T,H,W) in [-1,1]) or use --synthetic for seeded random
conditioning of the right shapes (benchmarks / smoke runs).  Outputs: `<save_path>/result/sample_XXXX.npy`
(frames in [0,1], (T,H,W,3)) + `log_info.json` with the resume-skip list, like the reference.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--seed", type=int, default=42)
    p.add_argument("--config_path", type=str, default="")
    p.add_argument("--ckpt_path", type=str, default="")
    p.add_argument("--vae_path", type=str, default="")
    p.add_argument("--cond_path", type=str, default="", help="precomputed conditioning tensors (see module docstring)")
    p.add_argument("--synthetic", action="store_true", help="seeded random conditioning + name-keyed synthetic weights")
    p.add_argument("--save_path", type=str, default="outputs/demo/tv2v")
    p.add_argument("--H", type=int, default=256)
    p.add_argument("--W", type=int, default=384)
    p.add_argument("--num_keyframes", type=int, default=9)
    p.add_argument("--prompt", type=str, default="")
    p.add_argument("--negative_prompt", type=str, default="ugly, low quality")
    p.add_argument("--add_prompt", type=str, default="masterpiece, high quality")
    p.add_argument("--sample_steps", type=int, default=50)
    p.add_argument("--sampler_name", type=str, default="DPMPP2SAncestralSampler")
    p.add_argument("--discretization_name", type=str, default="LegacyDDPMDiscretization")
    p.add_argument("--cfg_scale", type=float, default=7.5)
    p.add_argument("--num_samples", type=int, default=1)
    p.add_argument("--disable_check_repeat", action="store_true")
    return p.parse_args()


def init_sampling(name: str, steps: int, scale: float, discretization: str):
    """scripts/sampling/util.py:385-556 reduced to the samplers of the hot path (eta=1, s_noise=1, VanillaCFGTV2V)."""
    from ccedit_amd.config import instantiate_from_config
    dd = "sgm.modules.diffusionmodules."
    if name not in ("DPMPP2SAncestralSampler", "EulerAncestralSampler"):
        raise NotImplementedError(f"sampler {name}: only the ancestral samplers of the hot path are built")
    return instantiate_from_config(dict(target=dd + "sampling." + name, params=dict(
        num_steps=steps, eta=1.0, s_noise=1.0, verbose=True,
        discretization_config=dict(target=dd + "discretizer." + discretization),
        guider_config=dict(target=dd + "guiders.VanillaCFGTV2V", params=dict(scale=scale)))))


def main():
    args = parse()
    torch.manual_seed(args.seed)
    torch.set_grad_enabled(False)
    from ccedit_amd.config import instantiate_from_config, load_config
    from ccedit_amd.utils.synth import fill_module_
    if not args.config_path:
        raise SystemExit("--config_path is required (e.g. configs/inference_ccedit/keyframe_no2ndca_depthmidas.yaml)")
    cfg = load_config(args.config_path)
    dev = torch.device("cuda")
    with torch.device(dev):
        model = instantiate_from_config(cfg.model)
    if args.ckpt_path:
        model.init_from_ckpt(args.ckpt_path)
    elif args.synthetic:
        fill_module_(model.model, prefix="model.")
        fill_module_(model.first_stage_model, prefix="first_stage_model.")
    else:
        raise SystemExit("need --ckpt_path or --synthetic")
    model.pack(dev)

    T, h, w = args.num_keyframes, args.H // 8, args.W // 8
    g = torch.Generator().manual_seed(args.seed)
    if args.cond_path:
        if args.cond_path.endswith(".safetensors"):
            from safetensors.torch import load_file
            cond = load_file(args.cond_path)
        else:
            cond = torch.load(args.cond_path, map_location="cpu")
        cross_c, cross_uc, hint = cond["crossattn"], cond["crossattn_uc"], cond["control_hint"]
    else:
        cross_c, cross_uc = torch.randn(1, 77, 768, generator=g), torch.randn(1, 77, 768, generator=g)
        hint = (torch.rand(1, 1, T, args.H, args.W, generator=g) * 2 - 1).repeat(1, 3, 1, 1, 1)
    c = dict(crossattn=cross_c.to(dev), control_hint=hint.to(dev))
    uc = dict(crossattn=cross_uc.to(dev), control_hint=hint.clone().to(dev))     # uc keeps the SAME hint (:339-344)

    os.makedirs(os.path.join(args.save_path, "result"), exist_ok=True)
    log_path = os.path.join(args.save_path, "log_info.json")
    log = json.load(open(log_path)) if os.path.exists(log_path) else {"done": []}
    for i in range(args.num_samples):
        tag = f"sample_{i:04d}"
        if tag in log["done"] and not args.disable_check_repeat:
            continue
        randn = torch.randn(1, 4, T, h, w, generator=g).to(dev)                   # CPU generator, like :363
        sampler = init_sampling(args.sampler_name, args.sample_steps, args.cfg_scale, args.discretization_name)
        t0 = time.time()
        z = sampler(lambda inp, sigma, cc: model.denoiser(model.model, inp, sigma, cc), randn, c, uc=uc)
        x = model.decode_first_stage(z)
        torch.cuda.synchronize()
        x = torch.clamp((x + 1.0) / 2.0, 0.0, 1.0)                               # :473-475
        frames = x[0].permute(1, 2, 3, 0).cpu().numpy()
        import numpy as np
        np.save(os.path.join(args.save_path, "result", tag + ".npy"), frames)
        print(f"{tag}: {T} frames {args.H}x{args.W} in {time.time() - t0:.2f}s")
        log["done"].append(tag)
        json.dump(log, open(log_path, "w"))


if __name__ == "__main__":
    main()
