#!/usr/bin/env python3
"""TVI2V (reference-frame) sampling entry point — counterpart of the reference's
scripts/sampling/sampling_tv2v_ref.py: everything of sampling_tv2v.py plus the edited centre frame `cond_img`
((1,3,H,W) in [-1,1]; VAE-encoded to `cond_feat` by the conditioner's VAEEmbedder and, on the network side, fed to
controlnet_img and the anchor cross-frame attention) and `--prior_type {video, ref, video_ref}` for the noise prior
(sampling_tv2v_ref.py:415-437).  Config: configs/inference_ccedit/keyframe_ref_cp_no2ndca_add_cfca_depthzoe.yaml.
`--cond_path` additionally holds `cond_img`.  Job mode (lists, `--videos_directory`, `--json_path` + `--videos_root` + `--reference_root` with
one `output-<Target Prompt>.png` per job, `--batch_size`, `--auto_ref_editing`, which raises as in the reference): see sampling_tv2v.py.
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

from scripts.sampling.sampling_tv2v import (add_common_args, build_model, conditioning_tensors, job_mode, run_jobs, sample_one,  # noqa: E402
                                            save_result, text_inputs)


def main():
    p = argparse.ArgumentParser()
    add_common_args(p)
    p.add_argument("--prior_type", type=str, default="ref", choices=["video", "ref", "video_ref"])
    p.add_argument("--reference_path", type=str, default="", help="edited centre frame (image file) -> cond_img")
    p.add_argument("--reference_root", type=str, default="", help="path to the root of reference videos")
    p.add_argument("--auto_ref_editing", action="store_true", help="auto center editing")
    args = p.parse_args()
    torch.manual_seed(args.seed)
    torch.set_grad_enabled(False)
    if job_mode(args):        # the reference script's list / directory / BalanceCC-json surface (sampling_tv2v_ref.py:124-194, 340-400)
        return run_jobs(args, with_ref=True)
    if args.auto_ref_editing:
        print("Conduct auto ref editing, args.reference_path is ignored.")
        raise NotImplementedError          # as the reference (sampling_tv2v_ref.py:366-369)
    from scripts.sampling.util import ResumeLog
    model, dev = build_model(args)
    T, h, w = args.num_keyframes, args.H // 8, args.W // 8
    g = torch.Generator().manual_seed(args.seed)
    need_frames = (args.prior_coefficient_x != 0.0 and args.prior_type != "ref") or args.sdedit_denoise_strength != 0.0
    cond = conditioning_tensors(args, g, need_frames, need_ref=True)
    hint, ref = cond["control_hint"].to(dev), cond["cond_img"].to(dev)
    txt, txt_uc = text_inputs(cond, dev)
    batch = {"txt": txt, "control_hint": hint, "cond_img": ref}
    batch_uc = {"txt": txt_uc, "control_hint": hint.clone(), "cond_img": ref.clone()}
    c, uc = model.conditioner.get_unconditional_conditioning(batch, batch_uc=batch_uc)     # two VAE encodes (two RNG draws)
    keyframes = cond["keyframes"].to(dev) if need_frames else None
    log = ResumeLog(args.save_path)
    for i in range(args.num_samples):
        tag = f"sample_{i:04d}"
        if log.done(tag) and not args.disable_check_repeat:
            continue
        randn = torch.randn(1, 4, T, h, w, generator=g).to(dev)
        t0 = time.time()
        x = sample_one(args, model, dev, c, uc, randn, keyframes=keyframes, ref=ref, prior_type=args.prior_type)
        torch.cuda.synchronize()
        save_result(args, tag, x)
        print(f"{tag}: {T} frames {args.H}x{args.W} in {time.time() - t0:.2f}s")
        log.add(tag)


if __name__ == "__main__":
    main()
