#!/usr/bin/env python3
"""Headline benchmark: UNet denoising steps/s on synthetic 17x512x768 clips (BASELINE.json metric).

One "step" = one network evaluation of the TV2V hot path on the CFG-doubled batch
(OpenAIWrapperControlLDM3DTV2V.forward: hint remap -> ControlNet2D on 34 frames -> pseudo-3D UNet ->
eps), B=2 (uncond+cond), T=17 keyframes, latent 64x96, context 77x768, hint 3x512x768 per frame —
77.68 TFLOP algorithmic.  The DPMPP2SAncestral sampler calls exactly this 59 times per
30-step clip; after the timed steps one whole clip (sampler loop + AutoencoderKL decode) is timed as well and
its frames/s are added to the JSON line (`clip`; `--no-clip` skips it).

    python bench.py --gpus N --steps K --warmup W
    (N>1: python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...)

Multi-GPU: `--gpus N` without a launcher environment (WORLD_SIZE unset) re-executes itself under torch.distributed.run with N ranks
on 127.0.0.1, one per GPU.  `value` at N > 1 is BASELINE.json config 5 — N independent clips, one per GPU ("replicas only", no
data-path collective; barrier on both sides, MAX time over ranks), `scaling: "weak"`.  The same line carries a `c4` object: config 4,
ONE clip whose latent rows are sharded over the N ranks (parallel.RowShard: halo rows, all-reduced GroupNorm sums, head-parallel
spatial attention through all-to-alls; `--shard-mode` selects another decomposition) — strong-scaling ms/step,
`efficiency_per_gpu` against the single-GPU step measured in the same run, exchanges and bytes per step.  That section runs in CHILD
processes with a process group of their own (c4_in_children; `--c4-inline` = inside the benchmark processes under a watchdog): the
sharded path has never met a multi-GPU node, and neither a crash in the runtime nor a stuck collective there may cost the replica line.
`--shard-frames` makes config 4 the headline `value` instead (`scaling: "strong"`).

Prints ONE JSON line on rank 0.  Inputs are resident in HBM before the timed region.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FLOP_PER_STEP = 77.68e12          # SURVEY.md §8d, measured from the reference modules on the meta device
FLOP_PER_STEP_TVI2V = 110.31e12   # BASELINE.json config 3 (controlnet_img + anchor cross-frame attention)
MFMA_PEAK_TFLOPS = 2500.0         # MI355X dense bf16 (MI355X_MICROARCH.md)
HBM_PEAK_GBS = 8000.0             # HBM3E spec (ibid.; a float4 copy measures 6290)
T, H, W, L, CTX = 17, 64, 96, 77, 768


def synth_inputs(device, seed=42, b=1):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(b, 4, T, H, W, generator=g)
    cross_c = torch.randn(b, L, CTX, generator=g)
    cross_uc = torch.randn(b, L, CTX, generator=g)
    low = torch.rand(b, 1, T, H // 4, W // 4, generator=g)
    hint = torch.nn.functional.interpolate(low, size=(T, 8 * H, 8 * W), mode="trilinear", align_corners=False)
    hint = (hint * 2 - 1).repeat(1, 3, 1, 1, 1).contiguous()
    return x.to(device), cross_c.to(device), cross_uc.to(device), hint.to(device)


def build_model(device, tvi2v=False):
    from ccedit_amd.sgm_compat import build_network
    from ccedit_amd.utils.synth import fill_module_
    w = build_network(device, crossframe=tvi2v)    # full-size network, parameters created on the GPU
    fill_module_(w, prefix="model.")               # name-keyed synthetic weights (device generator)
    w.diffusion_model.pack(device)
    w.cache_hint_stem = False                      # the per-step metric recomputes the hint stem every step
    return w


def cpu_model() -> str:
    try:
        with open("/proc/cpuinfo") as f:
            for ln in f:
                if ln.startswith("model name"):
                    return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    import platform
    return platform.processor() or "unknown"


def cpu_baseline(wrapper, tvi2v=False, device=None, full_step=False):
    """Oracle (CPU fp32 restatement, `kind: port`) on a bounded sample of the same workload: the same
    full-width network and weights on a crop — B=2 (CFG), T=6 keyframes, latent 32x48 — timed on the host
    cores; converted to steps/s through the FLOPs ATen actually executed (FlopCounterMode).  The full-size step is NOT run on
    the CPU (about 100 s per evaluation): `value` is the crop's FLOP rate divided by the full-size step's FLOPs."""
    from oracle import ccedit_oracle as O
    from torch.utils.flop_counter import FlopCounterMode
    threads = min(os.cpu_count() or 1, 32)        # ATen's CPU kernels stop scaling (and collapse) far below 256 threads
    torch.set_num_threads(threads)
    sd = {"model." + k: v.detach().float().cpu() for k, v in wrapper.state_dict().items()}
    g = torch.Generator().manual_seed(7)
    tt, hh, ww = 6, 32, 48
    x = torch.randn(2, 4, tt, hh, ww, generator=g)
    c = dict(crossattn=torch.randn(2, L, CTX, generator=g), control_hint=torch.rand(2, 3, tt, 8 * hh, 8 * ww, generator=g) * 2 - 1)
    if tvi2v:
        cf = torch.randn(1, 4, hh, ww, generator=g) * 0.18215
        c["cond_feat"] = torch.cat([cf, cf])
    t = torch.tensor([601, 601], dtype=torch.int64)
    with torch.no_grad():
        with FlopCounterMode(display=False) as fc:
            t0 = time.time()
            O.network_forward(sd, O.NetConfig(crossframe=tvi2v), x, t, c)
            dt = time.time() - t0
    flops = float(fc.get_total_flops())
    per_step = FLOP_PER_STEP_TVI2V if tvi2v else FLOP_PER_STEP
    out = dict(value=(flops / dt) / per_step, unit="UNet steps/s (FLOP-equivalent)", cores=threads, cpu=cpu_model(), host_cpus=os.cpu_count(),
               kind="port", seconds=round(dt, 2), cpu_tflops=round(flops / dt / 1e12, 3),
               sample=f"oracle network_forward ({'TVI2V' if tvi2v else 'TV2V'}), full-width weights, B=2 T={tt} latent {hh}x{ww}: "
                      f"{flops/1e12:.2f} TFLOP in {dt:.1f}s; steps/s = CPU FLOP/s / {per_step / 1e12:.2f} TFLOP (a conversion of the "
                      f"crop's rate — the full-size step itself is not run on the CPU)")
    if full_step:
        out["full_size_step"] = cpu_full_size_step(sd, tvi2v)
    if not tvi2v:
        out.update(cpu_config1_end_to_end(sd, threads, wrapper, device))
    return out


def cpu_full_size_step(sd, tvi2v=False):
    """BASELINE.md section 3, second half (`--cpu-full-step`): ONE measured network evaluation of the oracle at the full 17 x 512 x 768
    size — the same CFG-doubled batch the GPU step evaluates (B = 2 x T = 17, latent 64 x 96; 77.68 / 110.31 TFLOP in fp32) — and the
    x (2 N - 1) extrapolation to a clip.  Not part of the default run (about a minute and a half of host time)."""
    from oracle import ccedit_oracle as O
    threads = min(os.cpu_count() or 1, 32)        # measured on the GPU box's 128-core host: 90 s on 16 or 32 threads, 113 s on 64, 180 s on 128
    torch.set_num_threads(threads)
    g = torch.Generator().manual_seed(42)
    x = torch.randn(1, 4, T, H, W, generator=g)
    cc, cu = torch.randn(1, L, CTX, generator=g), torch.randn(1, L, CTX, generator=g)
    hint = (torch.rand(1, 1, T, 8 * H, 8 * W, generator=g) * 2 - 1).repeat(1, 3, 1, 1, 1)
    c = dict(crossattn=torch.cat([cu, cc]), control_hint=torch.cat([hint, hint]))
    if tvi2v:
        cf = torch.randn(1, 4, H, W, generator=g) * 0.18215
        c["cond_feat"] = torch.cat([cf, cf])
    t = torch.tensor([601, 601], dtype=torch.int64)
    with torch.no_grad():
        t0 = time.time()
        eps = O.network_forward(sd, O.NetConfig(crossframe=tvi2v), torch.cat([x, x]), t, c)
        dt = time.time() - t0
    assert eps.shape == (2, 4, T, H, W) and bool(torch.isfinite(eps).all())
    per_step = FLOP_PER_STEP_TVI2V if tvi2v else FLOP_PER_STEP
    n_eval = 99 if tvi2v else 59
    return dict(seconds=round(dt, 1), steps_per_s=round(1.0 / dt, 5), cpu_tflops=round(per_step / dt / 1e12, 3), cores=threads,
                clip_extrapolated_s=round(n_eval * dt, 0), clip_evaluations=n_eval,
                workload=f"oracle network_forward ({'TVI2V' if tvi2v else 'TV2V'}) at the full size: B=2 x T={T}, latent {H}x{W}, "
                         f"{per_step / 1e12:.2f} TFLOP, fp32, {threads} threads — one measured evaluation; x{n_eval} = the sampler's evaluations of a clip")


def cpu_config1_end_to_end(sd, threads, wrapper=None, device=None):
    """BASELINE.md §3 protocol, first half: BASELINE.json config 1 END TO END on the oracle — 4 keyframes at 256x256 (latent
    32x32), 5 DPMPP2SAncestral steps at cfg 7.5 (9 network evaluations on the CFG-doubled batch) and the AutoencoderKL decode
    of the 4 frames, full-width network and VAE, same name-keyed synthetic weights as the GPU run.
    The HIP path then runs the SAME trajectory (same initial latent, conditioning and per-step ancestral noise) through the product's
    sampler / denoiser / guider / wrapper / VAE, and the line carries the distance of its final latent and decoded frames from the
    oracle's: shipped width, multi-step, end to end — the oracle here is the checker, not the thing measured."""
    from oracle import ccedit_oracle as O
    from ccedit_amd.sgm_compat import build_vae
    from ccedit_amd.utils.synth import fill_module_
    vae = build_vae(torch.device("cpu"))
    fill_module_(vae, prefix="first_stage_model.")
    vsd = {"first_stage_model." + k: v.detach().float() for k, v in vae.state_dict().items()}
    g = torch.Generator().manual_seed(11)
    tt, hh, ww = 4, 32, 32
    x = torch.randn(1, 4, tt, hh, ww, generator=g)
    hint = torch.rand(1, 3, tt, 8 * hh, 8 * ww, generator=g) * 2 - 1
    c = dict(crossattn=torch.randn(1, L, CTX, generator=g), control_hint=hint)
    uc = dict(crossattn=torch.randn(1, L, CTX, generator=g), control_hint=hint.clone())
    noises = [torch.randn(x.shape, generator=g) for _ in range(5)]          # one draw per sampler step, the last included
    table = O.denoiser_sigmas()
    evals = [0]

    def net(xx, idx, cond):
        evals[0] += 1
        return O.network_forward(sd, O.NetConfig(), xx, idx, cond)

    it = iter(noises)
    with torch.no_grad():
        t0 = time.time()
        z = O.dpmpp2s_ancestral_sample(lambda xx, sig, cond: O.discrete_denoise(net, table, xx, sig, cond), x, c, uc, 5, 7.5,
                                       lambda v: next(it))
        t1 = time.time()
        frames = O.vae_decode(vsd, "first_stage_model", O.VAEConfig(), z)
        t2 = time.time()
    assert frames.shape == (1, 3, tt, 8 * hh, 8 * ww) and bool(torch.isfinite(frames).all())
    out = dict(c1_end_to_end_s=round(t2 - t0, 2), c1_sampler_s=round(t1 - t0, 2), c1_vae_decode_s=round(t2 - t1, 2),
               c1_evaluations=evals[0], c1_frames_per_s=round(tt / (t2 - t0), 4),
               c1_workload=f"BASELINE config 1: {tt} keyframes 256x256, 5 DPMPP2SAncestral steps cfg 7.5 ({evals[0]} evaluations) + "
                           f"VAE decode, oracle fp32 on {threads} threads")
    if wrapper is not None and device is not None:
        out["c1_hip_vs_oracle"] = hip_config1(wrapper, device, x, c, uc, noises, z, frames, vsd)
    return out


def hip_config1(wrapper, device, x, c, uc, noises, z_ref, frames_ref, vsd):
    """The product path on config 1 with the oracle's inputs and noise; relative RMS distance of the final latent / frames."""
    from ccedit_amd.config import instantiate_from_config
    from ccedit_amd.sgm_compat import build_vae
    from ccedit_amd import ops
    dd = "sgm.modules.diffusionmodules."
    denoiser = instantiate_from_config(dict(target=dd + "denoiser.DiscreteDenoiser", params=dict(
        num_idx=1000, weighting_config=dict(target=dd + "denoiser_weighting.EpsWeighting"),
        scaling_config=dict(target=dd + "denoiser_scaling.EpsScaling"),
        discretization_config=dict(target=dd + "discretizer.LegacyDDPMDiscretization"))))
    sampler = instantiate_from_config(dict(target=dd + "sampling.DPMPP2SAncestralSampler", params=dict(
        num_steps=5, eta=1.0, s_noise=1.0, discretization_config=dict(target=dd + "discretizer.LegacyDDPMDiscretization"),
        guider_config=dict(target=dd + "guiders.VanillaCFGTV2V", params=dict(scale=7.5)))))
    it = iter([n.to(device) for n in noises])
    sampler.noise_sampler = lambda v: next(it)
    vae = build_vae(device)
    # the oracle's VAE weights (name-keyed fills differ between the CPU and the device generator: same tensors on both sides)
    missing = vae.load_state_dict({k[len("first_stage_model."):]: v for k, v in vsd.items()}, strict=False)
    assert not missing.missing_keys, missing.missing_keys[:4]
    vae.pack(device)
    cd = {k: v.to(device) for k, v in c.items()}
    ud = {k: v.to(device) for k, v in uc.items()}
    keep = wrapper.cache_hint_stem
    wrapper.cache_hint_stem = True
    t0 = time.perf_counter()
    z = sampler(lambda inp, sig, cc: denoiser(wrapper, inp, sig, cc), x.to(device).clone(), cd, uc=ud)
    zs = ops.axpby(z.contiguous(), z.contiguous(), 1.0 / 0.18215, 0.0)
    from ccedit_amd import policy
    vae.precision = "bf16" if policy.get("vae_fp32") == 0 else "fp32"       # the engine's default for the shipped yamls (time_clip)
    frames = vae.decode(zs)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    wrapper.cache_hint_stem = keep
    wrapper.reset_caches()

    def rel(a, b):
        a, b = a.double().cpu(), b.double()
        return float(((a - b) ** 2).mean().sqrt() / (b ** 2).mean().sqrt())

    r_lat, r_fr = rel(z, z_ref), rel(frames, frames_ref)
    # the decoder alone, in the reference's precision (policy vae_fp32): the ORACLE's final latent through the fp32 first stage
    vae.precision = "fp32"
    zr = z_ref.to(device=device, dtype=torch.float32).contiguous()
    r_dec32 = rel(vae.decode(ops.axpby(zr, zr, 1.0 / 0.18215, 0.0)), frames_ref)
    vae.precision = "bf16"
    r_dec16 = rel(vae.decode(ops.axpby(zr, zr, 1.0 / 0.18215, 0.0)), frames_ref)
    ok = bool(torch.isfinite(frames).all()) and r_lat < C1_LATENT_TOL and r_fr < C1_FRAMES_TOL and r_dec32 < C1_DECODE_FP32_TOL
    if not ok:
        sys.stderr.write(f"bench.py: config 1 on the HIP path is {r_lat:.4f} (latent) / {r_fr:.4f} (frames) from the oracle — outside the "
                         f"stated budget {C1_LATENT_TOL} / {C1_FRAMES_TOL}\n")
    return dict(final_latent_rel_rms=round(r_lat, 5), frames_rel_rms=round(r_fr, 5), within_budget=ok,
                budget=dict(latent=C1_LATENT_TOL, frames=C1_FRAMES_TOL, decode_fp32=C1_DECODE_FP32_TOL), hip_end_to_end_s=round(dt, 3),
                decode_of_oracle_latent_rel_rms=dict(fp32_vae=float(f"{r_dec32:.3g}"), bf16_vae=float(f"{r_dec16:.3g}")),
                note="same initial latent, conditioning and per-step ancestral noise as the oracle run; bf16 HIP network vs fp32 oracle over "
                     "9 evaluations at the shipped width, then the VAE decode in the product's default precision (fp32 unless policy vae_fp32=0)")


# Drift budget of the 5-step config-1 trajectory (bf16 path against the fp32 oracle): one evaluation is held to 5e-2 relative RMS
# (SURVEY 8d); ancestral noise re-injection keeps the per-step errors from compounding, tests/test_network_gpu.py holds the
# reduced-width trajectory to 8e-2.
C1_LATENT_TOL, C1_FRAMES_TOL = 8e-2, 8e-2
C1_DECODE_FP32_TOL = 2e-5      # the fp32 first stage (policy vae_fp32) on the oracle's own final latent: fp32 rounding only


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--clip", action="store_true", help="(default on) time one full 30-step clip + VAE decode (frames/s)")
    ap.add_argument("--no-clip", action="store_true", help="skip the whole-clip timing (30-step sampler + VAE decode, ~7 s)")
    ap.add_argument("--json-out", type=str, default="", help="also write the JSON line to this file (rank 0)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-full-step", action="store_true",
                    help="cpu_baseline: also time ONE full-size (17x512x768, CFG-doubled) evaluation of the oracle on the host (BASELINE.md section 3; ~2 min)")
    ap.add_argument("--shard-frames", action="store_true",
                    help="N>1: BASELINE.json config 4 — ONE clip, its T=17 keyframes sharded over the ranks; default N>1 mode "
                         "is config 5 (one clip per GPU, no collective)")
    ap.add_argument("--shard-mode", choices=["pair", "a2a", "halo", "rows", "rows-pair"], default="rows",
                    help="pair: all-to-all layout transposition around the temporal ops, the two CFG halves on mirrored "
                         "partitions, two communicators and two streams; a2a: the same transposition, one partition; halo: "
                         "round-1 halo p2p + statistics all-reduce + K/V all-gather; rows: every rank holds all keyframes of 1/N of the "
                         "latent rows (halo rows for the 3x3 convs, all-reduced GroupNorm sums, all-gathered K/V, temporal ops local)")
    ap.add_argument("--workload", choices=["tv2v", "tvi2v"], default="tv2v",
                    help="tv2v = BASELINE.json config 2 (the headline metric); tvi2v = config 3 (ref-frame cfca network)")
    ap.add_argument("--no-profile-step", action="store_true", help="skip the extra HIP-event profiled step (PMC runs)")
    ap.add_argument("--no-tvi2v", action="store_true", help="skip the config-3 (TVI2V) step timing added to the default single-GPU line")
    ap.add_argument("--no-c4", action="store_true", help="N>1: skip the config-4 (one clip, rows sharded) object of the replica line")
    ap.add_argument("--c4-deadline", type=float, default=240.0, help="N>1: seconds the config-4 section may take before the line is emitted without it")
    ap.add_argument("--c4-inline", action="store_true", help="N>1: run the config-4 section inside the benchmark processes (the round-5 form) instead of "
                                                            "in child processes of their own")
    ap.add_argument("--c4-child", action="store_true", help=argparse.SUPPRESS)          # internal: this process IS a config-4 child (see c4_in_children)
    ap.add_argument("--c4-single-ms", type=float, default=0.0, help=argparse.SUPPRESS)  # internal: the parent's replica ms/step, for efficiency_per_gpu
    ap.add_argument("--attn", choices=["heads", "gather"], default="heads",
                    help="row-sharded spatial attention: all-to-all by head (heads) or all-gather of K/V (gather)")
    ap.add_argument("--dump-shapes", type=str, default="", help="write the GEMM launch shape sequence of one step (json)")
    ap.add_argument("--breakdown", action="store_true", help="print per-shape GEMM / attention time of one step to stderr")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # started like the single-GPU run (`python bench.py --gpus N`): become the launcher — one rank per GPU over RCCL on 127.0.0.1
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    # CCEDIT_DIST_BACKEND=gloo: development only — lets N ranks share the GPUs of a smaller box (ranks wrap around the
    # visible devices, collectives stage through host memory).  The driver's runs use the default: RCCL, one GPU per rank.
    backend = os.environ.get("CCEDIT_DIST_BACKEND", "nccl")
    if backend != "nccl":
        local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist_mod
        dist = dist_mod
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"

    from ccedit_amd import hip, ops
    hip.lib()                                       # fail loudly if the HIP library is missing
    torch.set_grad_enabled(False)
    tvi2v = args.workload == "tvi2v"
    flop_per_step = FLOP_PER_STEP_TVI2V if tvi2v else FLOP_PER_STEP
    wrapper = build_model(device, tvi2v)
    shard = args.shard_frames and world > 1
    x, cross_c, cross_uc, hint = synth_inputs(device, seed=42 + (0 if shard else rank))
    shards = install_shards(wrapper, args) if shard else ()
    inp = {}

    def set_inputs(x, cross_c, cross_uc, hint, seed):
        inp["x2"] = torch.cat([x, x]).contiguous()             # CFG-doubled batch, uc first (guiders.py:63)
        inp["cond"] = dict(crossattn=torch.cat([cross_uc, cross_c]).contiguous(), control_hint=torch.cat([hint, hint]).contiguous())
        if tvi2v:
            cf = (torch.randn(1, 4, H, W, generator=torch.Generator().manual_seed(seed)) * 0.18215).to(device)
            inp["cond"]["cond_feat"] = torch.cat([cf, cf]).contiguous()

    set_inputs(x, cross_c, cross_uc, hint, 7 + (0 if shard else rank))
    tstep = torch.tensor([601, 601], dtype=torch.int64, device=device)

    def step():
        return wrapper(inp["x2"], tstep, inp["cond"])

    def timed(n_steps):
        """W warm-up steps are the caller's; K steps between barrier + synchronize on both sides, MAX over ranks (seconds)."""
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n_steps):
            o = step()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if dist is not None:
            dt = max_over_ranks_ms(dt, dist, device, backend)
        return dt, o

    if args.c4_child:
        # config 4 in a process of its own (c4_in_children): ONE clip row-sharded over the ranks, the same clip on all ranks
        set_inputs(*synth_inputs(device, seed=42), 7)
        shards = install_shards(wrapper, args)
        step(); step()
        c4_dt, o4 = timed(args.steps)
        assert torch.isfinite(o4).all()
        info = shard_counters(shards, step, world, dist, wrapper)
        if rank == 0:
            print("C4_OBJECT " + json.dumps(c4_object(args, world, c4_dt / args.steps * 1e3, args.c4_single_ms, info)), flush=True)
        dist.barrier()
        dist.destroy_process_group()
        return

    # one-time initialisation, like building the model: the wrapper runs its first evaluation eagerly and captures the second into
    # a HIP graph (ccedit_amd/network.py) — with fewer than two warm-up steps that capture would fall into the timed region
    for _ in range(max(0, 2 - args.warmup)):
        step()
    for _ in range(args.warmup):
        step()
    dt, out = timed(args.steps)
    assert torch.isfinite(out).all()
    ms_per_step = dt / args.steps * 1e3
    value = (1 if shard else world) * args.steps / dt

    # ---- config 4 (ONE clip sharded over the ranks): one instrumented step (bytes, exchange count, device time inside exchanges) and
    # the single-GPU time the per-GPU efficiency is quoted against.  Headline mode (--shard-frames): the unsharded step is timed on
    # every rank afterwards.  Replica mode (default N > 1): the replica step above IS the single-GPU step; the sharded one is timed
    # here, on the same clip on all ranks, and reported as the `c4` object next to the replica `value`. ----
    roof = None
    extra = {}
    if shard:
        info = shard_counters(shards, step, world, dist, wrapper)
        keep, wrapper.frame_shard = wrapper.frame_shard, None
        keep_rows, wrapper.row_shard = wrapper.row_shard, None
        step(); step()
        single_ms = timed(args.steps)[0] / args.steps * 1e3
        wrapper.frame_shard, wrapper.row_shard = keep, keep_rows
        extra["shard"] = extra["c4"] = c4_object(args, world, ms_per_step, single_ms, info)
    if rank == 0 and args.dump_shapes and not shard:
        ops.PROFILE = ops.LaunchProfile()
        step()
        torch.cuda.synchronize()
        with open(args.dump_shapes, "w") as f:
            json.dump([list(map(str, r[4])) + [r[2]] for r in ops.PROFILE.records["tap_gemm"]], f)
        ops.PROFILE = None
    if shard and rank != 0 and not args.no_profile_step:
        step()                                      # rank 0's profiled step below still needs its collective partners
    if rank == 0 and not args.no_profile_step:
        ops.PROFILE = ops.LaunchProfile()
        step()
        prof = ops.PROFILE.summary()
        if args.breakdown:
            for fam in ("tap_gemm", "attention"):
                for shape, n, ms, tf in ops.PROFILE.by_shape(fam, with_kernel=True)[:int(os.environ.get('CCEDIT_BREAKDOWN_ROWS', '40'))]:
                    print(f"{fam:9s} {str(shape[0]):52s} x{n:3d} {ms:8.3f} ms {tf:7.1f} TF/s  {shape[1]}", file=sys.stderr)
            mem = {}
            for a_, b_, _, nb, shape, _k, _ex in ops.PROFILE.records.get("memory", []):
                e = mem.setdefault(shape, [0, 0.0, 0.0])
                e[0] += 1; e[1] += a_.elapsed_time(b_); e[2] += nb
            for shape, (n, ms, nb) in sorted(mem.items(), key=lambda kv: -kv[1][1]):
                print(f"memory    {str(shape):60s} x{n:3d} {ms:8.3f} ms {nb / (ms * 1e-3) / 1e9:7.0f} GB/s", file=sys.stderr)
        by_kernel = []
        balance = MFMA_PEAK_TFLOPS * 1e12 / (HBM_PEAK_GBS * 1e9)       # 312 FLOP per byte: below it the HBM roof is the lower one
        for r in ops.PROFILE.by_kernel():
            row = dict(kernel=r["kernel"], launches=r["launches"], ms=round(r["ms"], 3), alg_bytes_per_launch=round(r["bytes"] / r["launches"]))
            flops = r["tflops"] * 1e12 * r["ms"] * 1e-3
            # which roof bounds a template: its algorithmic intensity (FLOPs / algorithmic bytes over all its launches) against the
            # machine balance.  Norm / concat / layout passes have no FLOPs; the streaming K = 320 Linears, the 320-channel temporal
            # convs, the few-channel hint-stem convs and the T = 17 temporal attention sit below the balance point too.
            if r["family"] == "memory" or flops / max(r["bytes"], 1.0) < balance:
                row.update(bound="hbm", gbytes_per_s=round(r["gbytes_per_s"], 1), frac=round(r["gbytes_per_s"] / HBM_PEAK_GBS, 4),
                           gbytes=round(r["bytes"] / 1e9, 3))
                if r["family"] != "memory":
                    row.update(tflops=round(r["tflops"], 1), flop_per_byte=round(flops / max(r["bytes"], 1.0), 1))
            else:
                # frac: the FLOPs the matrix pipe retired in this template's launches (executed) over the peak; the algorithmic
                # count of the reference's operations (upsample + conv as nine taps) beside it where the two differ
                row.update(bound="mfma", tflops=round(r["exec_tflops"], 1), frac=round(r["exec_tflops"] / MFMA_PEAK_TFLOPS, 4))
                if abs(r["tflops"] - r["exec_tflops"]) > 1e-6 * r["tflops"]:
                    row.update(algorithmic_tflops=round(r["tflops"], 1), algorithmic_frac=round(r["tflops"] / MFMA_PEAK_TFLOPS, 4))
            row["_ms"], row["_flops"], row["_bytes"] = r["ms"], r["exec_tflops"] * 1e12 * r["ms"] * 1e-3, r["bytes"]
            by_kernel.append(row)
        executed_flops = ops.PROFILE.executed
        ops.PROFILE = None
        g = prof["tap_gemm"]
        ach = g["flops"] / (g["total_ms"] * 1e-3) / 1e12
        # The DOMINANT kernel = the kernel TEMPLATE (by name: all instantiations and block shapes of `g8_kernel`, of `conv_halo_kernel`,
        # ... summed) with the most time in it over ALL rows — GEMM, attention and memory passes (split by
        # instantiation, the persistent GEMM's 34 ms hid behind the 13 ms of the single conv_halo symbol).  `achieved` / `frac` are
        # its EXECUTED FLOPs over its HIP-event time in this profiled step (reproducible from the rocprofv3 summary under profiles/:
        # sum over the template's rows of calls x average duration); `symbol_leader` is the single by_kernel row with the most time
        # (what earlier rounds reported as the dominant kernel).  The whole GEMM family stays as `family_*`.
        tmpl = {}
        for r in by_kernel:
            t = tmpl.setdefault(r["kernel"].split(" ")[0], dict(ms=0.0, flops=0.0, bytes=0.0, launches=0, rows=0))
            t["ms"] += r["_ms"]; t["flops"] += r["_flops"]; t["bytes"] += r["_bytes"]; t["launches"] += r["launches"]; t["rows"] += 1
        for r in by_kernel:
            for k_ in ("_ms", "_flops", "_bytes"):
                r.pop(k_)
        dname, dt_ = max(tmpl.items(), key=lambda kv: kv[1]["ms"])
        d_tf, d_gb = dt_["flops"] / (dt_["ms"] * 1e-3) / 1e12, dt_["bytes"] / (dt_["ms"] * 1e-3) / 1e9
        dom_mfma = dt_["flops"] / max(dt_["bytes"], 1.0) >= balance
        lead = max(by_kernel, key=lambda r: r["ms"])
        dom = dict(kernel=dname, launches=dt_["launches"], ms=round(dt_["ms"], 3), alg_bytes_per_launch=round(dt_["bytes"] / dt_["launches"]))
        roof = dict(bound="mfma" if dom_mfma else "hbm", kernel=dname, kernel_instantiations=dt_["rows"], kernel_launches=dt_["launches"],
                    kernel_ms_per_step=dom["ms"], avg_launch_us=round(1e3 * dt_["ms"] / dt_["launches"], 2),
                    achieved=round(d_tf if dom_mfma else d_gb, 1),
                    peak=MFMA_PEAK_TFLOPS if dom_mfma else HBM_PEAK_GBS, unit="TFLOP/s" if dom_mfma else "GB/s",
                    frac=round((d_tf / MFMA_PEAK_TFLOPS) if dom_mfma else (d_gb / HBM_PEAK_GBS), 4),
                    frac_basis="executed FLOPs (what the matrix pipe retired) / HIP-event time of the template's launches / dense bf16 peak",
                    traffic=None, algorithmic_bytes_per_launch=dom["alg_bytes_per_launch"],
                    symbol_leader=dict(kernel=lead["kernel"], launches=lead["launches"], ms_per_step=lead["ms"], frac=lead["frac"],
                                       bound=lead["bound"]),
                    by_template=[dict(kernel=k_, ms=round(v_["ms"], 3), launches=v_["launches"],
                                      tflops=round(v_["flops"] / (v_["ms"] * 1e-3) / 1e12, 1), gbytes_per_s=round(v_["bytes"] / (v_["ms"] * 1e-3) / 1e9, 1))
                                 for k_, v_ in sorted(tmpl.items(), key=lambda kv: -kv[1]["ms"])][:12],
                    family="ccedit_gemm + ccedit_ff320 (g8_kernel, conv_halo_kernel, tap_gemm_kernel, lin320s_kernel / lin320_kernel, lin640s_kernel, temp320s_kernel, ff320_kernel, small_conv3x3_kernel)",
                    family_achieved=round(ach, 1), family_frac=round(ach / MFMA_PEAK_TFLOPS, 4), family_launches=g["launches"],
                    family_avg_launch_us=round(g["avg_us"], 2), algorithmic_flops_per_step=g["flops"], by_kernel=by_kernel)
        # HBM traffic cannot be read from inside the process: it comes from the committed rocprofv3 PMC passes of this
        # same workload (tools/pmc_traffic.sh: FETCH_SIZE and WRITE_SIZE in separate runs, FETCH x2 per
        # MI355X_MICROARCH.md), averaged per tap_gemm launch like `achieved`.
        # The capture records the hash of the kernel sources it was taken from (tools/pmc_traffic.sh); a capture of OTHER
        # kernels is not reported: traffic stays null and the line says why.
        roof["family_algorithmic_bytes_per_launch"] = round(g["bytes"] / g["launches"])
        PMC_TRAFFIC_FILE = pmc_traffic_file(tvi2v)
        pmc = os.path.join(ROOT, "profiles", PMC_TRAFFIC_FILE)
        if os.path.exists(pmc):
            with open(pmc) as f:
                cap = json.load(f)
            tg = cap.get("tap_gemm")
            src = cap.get("kernel_source_hash")
            if src != kernel_source_hash():
                roof["traffic_note"] = (f"profiles/{PMC_TRAFFIC_FILE} was captured from kernel sources {src}, this build is "
                                        f"{kernel_source_hash()}: stale, not reported (re-run tools/pmc_traffic.sh)")
            else:
                sym = (cap.get("by_symbol") or {}).get(dom["kernel"].split(" ")[0])          # e.g. "attn_spatial_kernel"
                if sym and sym["launches"]:
                    roof["traffic"] = round((sym["fetch_bytes_x2"] + sym["write_bytes"]) / sym["launches"])
                    roof["traffic_unit"] = (f"bytes past the L2 per {dom['kernel'].split(' ')[0]} launch (rocprofv3 PMC: FETCH_SIZE x2 + "
                                            f"WRITE_SIZE, all instantiations of the template, profiles/{PMC_TRAFFIC_FILE})")
                    roof["traffic_launches"] = sym["launches"]
                if tg and tg["launches"]:
                    roof["family_traffic"] = round((tg["fetch_bytes_x2"] + tg["write_bytes"]) / tg["launches"])
                    roof["family_traffic_unit"] = "bytes past the L2 per ccedit_gemm / ccedit_ff320 launch"
                roof["traffic_kernel_source_hash"] = src
        a = prof.get("attention")
        if a:
            extra["attention"] = dict(tflops=round(a["flops"] / (a["total_ms"] * 1e-3) / 1e12, 1), launches=a["launches"],
                                      total_ms=round(a["total_ms"], 2), algorithmic_flops_per_step=a["flops"])
        extra["gemm_total_ms"] = round(g["total_ms"], 2)
        # what the matrix pipe really retires per step: the parity form of upsample + conv executes 4/9 of those convolutions'
        # multiply-adds, the hint stem runs on ONE of the two identical CFG halves and their shared prefix is evaluated once.  The
        # step's fraction of the MFMA peak is priced with THOSE FLOPs; `value` (steps/s) needs no FLOP count, and
        # the figure with the algorithmic count of the reference's operations (77.68 TFLOP) is kept beside it
        extra["executed_flops_per_step"] = executed_flops
        extra["step_tflops"] = round(executed_flops / (ms_per_step * 1e-3) / 1e12, 1)
        extra["step_frac_of_mfma_peak"] = round(executed_flops / (ms_per_step * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS, 4)
        extra["executed_frac_of_mfma_peak"] = extra["step_frac_of_mfma_peak"]          # (the name of rounds 4-5)
        extra["algorithmic_step_tflops"] = round(flop_per_step / (ms_per_step * 1e-3) / 1e12, 1)
        extra["algorithmic_step_frac_of_mfma_peak"] = round(flop_per_step / (ms_per_step * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS, 4)

    # BASELINE.json's metric names frames/s next to UNet steps/s: one whole clip (59 evaluations + sampler math + VAE decode)
    # outside the timed region above.  N > 1 replicas: every rank runs its own clip at the same time (rank 0's is reported);
    # frame-sharded: the one clip runs through the sharded wrapper on all ranks.
    clip = None
    if not args.no_clip:
        if dist is not None:
            dist.barrier()
        # config 3 (reference README.md:63-77): 50 steps = 99 evaluations at cfg 7 with the reference-frame latent as cond_feat
        clip = time_clip(wrapper, device, seed=43 if shard else 43 + rank, fp32_vae=world == 1,
                         **(dict(num_steps=50, scale=7.0, tvi2v=True) if tvi2v else {}))
        if rank != 0:
            clip = None

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(wrapper, tvi2v, device, full_step=args.cpu_full_step)

    # BASELINE.json config 3 on the default line: the TVI2V network's step, timed like the headline
    if rank == 0 and world == 1 and not tvi2v and not args.no_tvi2v:
        extra["tvi2v"] = time_tvi2v_step(device, args.steps)       # (a second full network beside the first: 288 GB of HBM)

    def emit():
        """Rank 0 prints the ONE JSON line (with whatever `extra` holds by now)."""
        if rank != 0:
            return
        line = {
            "metric": f"UNet denoising steps/s ({'TVI2V' if tvi2v else 'TV2V'} 17x512x768, bf16, CFG-doubled batch)", "value": round(value, 4),
            "unit": "UNet steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "strong" if shard else "weak",
            "vs_baseline": None,
            "dtype": "bf16", "data": "synthetic (seeded latent/context/depth hint; name-keyed random-init weights)",
            "config": {"workload": ("TVI2V ref-frame (cfca) + depth, 17x512x768, one network evaluation = ControlNet2D + controlnet_img + "
                                    "pseudo-3D UNet with anchor cross-frame attention on B=2 x T=17 frames; 110.31 TFLOP/step"
                                    if tvi2v else
                                    "TV2V depth-midas, 17x512x768, one network evaluation = ControlNet2D + pseudo-3D UNet on "
                                    "B=2 (cfg 7.5 uncond+cond) x T=17 frames, latent 64x96, 77x768 text context; "
                                    "77.68 TFLOP/step; a 30-step DPMPP2SAncestral clip = 59 such steps + VAE decode"),
                       "parallelism": ("one clip, the latent ROWS of every keyframe sharded over the ranks (halo rows for the 3x3 convs with the GroupNorm "
                                       "partial sums riding on them, " + ("head-parallel spatial attention through two all-to-alls" if args.attn == "heads"
                                                                          else "RCCL all-gather of K/V at the spatial attention") + ", temporal ops local"
                                       + ("; the CFG halves on two streams with a communicator each)" if args.shard_mode == "rows-pair" else ")")
                                       if shard and args.shard_mode.startswith("rows") else
                                       f"one clip, T=17 keyframes sharded over the ranks (mode {args.shard_mode}: " +
                                       ("halo p2p + stats all-reduce + K/V all-gather)" if args.shard_mode == "halo" else
                                        "all-to-all frame<->pixel transposition around every temporal op)") if shard else
                                       "1 clip per GPU (replicas, no collective)" if world > 1 else "single GPU"),
                       "hint_stem": "recomputed every step"},
            "roofline": roof, "cpu_baseline": cpu,
        }
        line.update(extra)
        # how many DEVICES the ranks really ran on: development runs (CCEDIT_DIST_BACKEND=gloo) let N ranks time-slice fewer GPUs — such a
        # line is a functional record, not a scaling measurement
        line["gpus_physical"] = min(world, torch.cuda.device_count())
        line["dist_backend"] = backend if world > 1 else None
        from ccedit_amd import policy
        line["policy_non_default"] = policy.non_default()
        if clip:
            line["clip"] = clip
        print(json.dumps(line))
        if args.json_out:            # the same line as a file that holds nothing else (stdout of a multi-rank run also carries the backends' banners)
            with open(args.json_out, "w") as f:
                json.dump(line, f)
                f.write("\n")


    # ---- config 4 next to the replica value (default N > 1): ONE clip row-sharded over the ranks, timed LAST and under a watchdog — this
    # path has never met a multi-GPU node, and a stuck collective must not cost the replica line the driver's scaling
    # curve is computed from: if the sharded section does not finish in time every rank emits / exits without it.
    if world > 1 and not shard and not args.no_c4 and not args.c4_inline:
        try:
            extra["c4"] = c4_in_children(args, rank, world, dist, ms_per_step)
        except Exception as e:            # (the port broadcast: the only collective of the parents in this section)
            extra["c4"] = dict(error=f"{type(e).__name__}: {e}")
    elif world > 1 and not shard and not args.no_c4:
        import threading

        def bail():
            extra["c4"] = dict(error=f"the row-sharded section did not finish within {args.c4_deadline} s; replica value unaffected")
            emit()
            sys.stdout.flush()
            os._exit(0)
        dog = threading.Timer(args.c4_deadline, bail)
        dog.daemon = True
        dog.start()
        try:
            xs = synth_inputs(device, seed=42)
            saved = dict(inp)
            set_inputs(*xs, 7)
            shards = install_shards(wrapper, args)
            step(); step()
            c4_dt, o4 = timed(args.steps)
            assert torch.isfinite(o4).all()
            info = shard_counters(shards, step, world, dist, wrapper)
            wrapper.frame_shard = wrapper.row_shard = None
            inp.update(saved)
            extra["c4"] = c4_object(args, world, c4_dt / args.steps * 1e3, ms_per_step, info)
        except Exception as e:            # an error on this rank: say so on the line instead of losing it
            extra["c4"] = dict(error=f"{type(e).__name__}: {e}")
        dog.cancel()
    emit()
    if dist is not None:
        dist.barrier()                              # rank 0 may still have been profiling: leave together
        dist.destroy_process_group()


def c4_in_children(args, rank, world, dist, single_ms):
    """Config 4 next to the replica value, ISOLATED: every rank starts a child process (this file with --c4-child) and the children form a
    process group of their own on another port, build the model again, time the row-sharded clip and hand the `c4` object to rank 0's
    parent through stdout.  The sharded path has never met a multi-GPU node: a crash inside the runtime (capturing RCCL
    collectives has produced those) or a stuck collective then costs the `c4` object — an `error` entry — and not the replica line the
    driver's scaling curve is computed from.  The parents keep their GPUs' memory (a second model fits many times) and wait."""
    import socket
    import subprocess
    port = [0]
    if rank == 0:
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port[0] = sk.getsockname()[1]
    dist.broadcast_object_list(port, src=0)
    env = dict(os.environ, MASTER_PORT=str(port[0]), MASTER_ADDR="127.0.0.1")
    env.pop("TORCHELASTIC_USE_AGENT_STORE", None)           # the children's rank 0 hosts the store of their own group
    cmd = [sys.executable, os.path.abspath(__file__), "--gpus", str(world), "--steps", str(args.steps), "--warmup", str(args.warmup), "--c4-child",
           "--c4-single-ms", f"{single_ms:.4f}", "--shard-mode", args.shard_mode, "--attn", args.attn, "--workload", args.workload]
    res = None
    try:
        out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=args.c4_deadline)
        if rank == 0:
            for ln in out.stdout.splitlines():
                if ln.startswith("C4_OBJECT "):
                    res = json.loads(ln[len("C4_OBJECT "):])
                    res["isolation"] = "child processes with a process group of their own"
            if res is None:
                res = dict(error=f"the config-4 child of rank 0 exited with code {out.returncode} and no result; replica value unaffected",
                           stderr_tail=out.stderr[-600:])
    except subprocess.TimeoutExpired:
        res = dict(error=f"the row-sharded section (child processes) did not finish within {args.c4_deadline} s; replica value unaffected")
    except Exception as e:                   # (spawn failure etc.)
        res = dict(error=f"{type(e).__name__}: {e}")
    return res if rank == 0 else None


def install_shards(wrapper, args):
    """BASELINE.json config 4: put the wrapper into one of the single-clip decompositions (ccedit_amd/parallel.py)."""
    from ccedit_amd.parallel import FrameShard, RowShard
    if args.shard_mode == "rows":               # the balanced decomposition (ceiling 1.0): rows of every frame, not keyframes
        shards = (RowShard(attn=args.attn),)
        wrapper.row_shard = shards[0]
    elif args.shard_mode == "rows-pair":        # ... with the CFG halves on two streams, a communicator each (exchanges of one half under
        shards = RowShard.cfg_pair(attn=args.attn)      # the other half's kernels); eager: two-stream capture of RCCL crashes the runtime
        wrapper.row_shard = shards
    elif args.shard_mode == "pair":             # a communicator per CFG half: their exchanges run independently
        shards = FrameShard.cfg_pair(T)
        wrapper.frame_shard = shards
    else:
        shards = (FrameShard(T, mode=args.shard_mode),)
        wrapper.frame_shard = shards[0]
    return shards


def shard_counters(shards, step, world, dist, wrapper=None):
    """One instrumented sharded step: per rank (bytes sent, exchanges, device ms inside exchanges, local keyframes).
    The counters advance in the Python exchange code, which a replayed HIP graph does not run (with RCCL the sharded evaluation is
    captured): the instrumented step is evaluated EAGERLY (ADVICE r5 — the c4 object reported 0 exchanges on real multi-GPU runs)."""
    for s_ in shards:
        s_.timing = []
        s_.reset_counters()
    saved = None if wrapper is None else wrapper.use_graph
    if wrapper is not None:
        wrapper.use_graph = False
    try:
        step()
        torch.cuda.synchronize()
    finally:
        if wrapper is not None:
            wrapper.use_graph = saved
    mine = torch.tensor([sum(s_.bytes_sent for s_ in shards), sum(s_.n_collectives for s_ in shards),
                         sum(s_.comm_ms() for s_ in shards), sum(getattr(s_, "t_local", T) for s_ in shards)], dtype=torch.float64)
    for s_ in shards:
        s_.timing = None
    allr = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather_object(allr, mine)
    return allr


def c4_object(args, world, sharded_ms, single_ms, allr):
    from ccedit_amd.parallel import cfg_pair_efficiency, row_sharding_efficiency, sharding_efficiency
    ceiling = (row_sharding_efficiency(H, world) if args.shard_mode.startswith("rows") else
               cfg_pair_efficiency(T, world) if args.shard_mode == "pair" else sharding_efficiency(T, world))
    return dict(
        config="BASELINE config 4: ONE 17x512x768 clip sharded over the ranks", mode=args.shard_mode,
        attention=(args.attn if args.shard_mode.startswith("rows") else None), scaling="strong",
        ms_per_step=round(sharded_ms, 3), steps_per_s=round(1e3 / sharded_ms, 4),
        frame_instances_per_rank=[int(a[3]) * (1 if args.shard_mode == "pair" else 2) for a in allr],
        latent_rows_per_rank=(H // world if args.shard_mode.startswith("rows") else H),
        measured_on="NOT measured over xGMI unless n_gpus real devices ran it",
        ceiling=round(ceiling, 4), single_gpu_ms_per_step=round(single_ms, 3),
        efficiency_per_gpu=round(single_ms / (world * sharded_ms), 4),
        exchanges_per_step=int(allr[0][1]), bytes_sent_per_step_max_rank=int(max(a[0] for a in allr)),
        exchange_ms_per_step_max_rank=round(max(float(a[2]) for a in allr), 3),
        exchange_ms_note="sum of device time between issue and completion of every exchange on its stream: an upper bound on "
                         "exposed communication")


def max_over_ranks_ms(ms, dist, device, backend):
    tt = torch.tensor([ms], dtype=torch.float64, device=device if backend == "nccl" else "cpu")
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    return float(tt.item())


def pmc_traffic_file(tvi2v: bool = False) -> str:
    """The committed rocprofv3 PMC capture (FETCH_SIZE / WRITE_SIZE passes, tools/pmc_traffic.sh) to read `traffic` from: the
    newest profiles/rNN_pmc_traffic.json taken from THIS build's kernel sources, else the newest one (reported as stale)."""
    import glob
    suffix = "_pmc_traffic_tvi2v.json" if tvi2v else "_pmc_traffic.json"
    names = sorted((os.path.basename(p) for p in glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]" + suffix))), reverse=True)
    for name in names:
        try:
            with open(os.path.join(ROOT, "profiles", name)) as f:
                if json.load(f).get("kernel_source_hash") == kernel_source_hash():
                    return name
        except (OSError, ValueError):
            continue
    return names[0] if names else "r06" + suffix


def kernel_source_hash() -> str:
    """sha256 over the kernel sources: ties a committed PMC capture to the code it was taken from."""
    import hashlib
    d = os.path.join(ROOT, "ccedit_amd", "csrc")
    h = hashlib.sha256()
    for name in sorted(os.listdir(d)):
        if name.endswith((".hip", ".h", ".cpp")):
            with open(os.path.join(d, name), "rb") as f:
                h.update(name.encode() + b"\0" + f.read())
    return h.hexdigest()[:16]


def time_tvi2v_step(device, steps):
    """BASELINE.json config 3: one evaluation of the TVI2V network (controlnet_img + anchor cross-frame attention, 110.31 TFLOP) on
    the same synthetic clip, 2 warm-up steps (eager, capture) + `steps` timed."""
    w = build_model(device, tvi2v=True)
    x, cross_c, cross_uc, hint = synth_inputs(device, seed=42)
    cf = (torch.randn(1, 4, H, W, generator=torch.Generator().manual_seed(7)) * 0.18215).to(device)
    cond = dict(crossattn=torch.cat([cross_uc, cross_c]).contiguous(), control_hint=torch.cat([hint, hint]).contiguous(),
                cond_feat=torch.cat([cf, cf]).contiguous())
    x2 = torch.cat([x, x]).contiguous()
    ts = torch.tensor([601, 601], dtype=torch.int64, device=device)
    for _ in range(3):
        o = w(x2, ts, cond)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        o = w(x2, ts, cond)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    assert torch.isfinite(o).all()
    return dict(config="BASELINE config 3: TVI2V ref-frame (cfca) + depth, 17x512x768, 110.31 TFLOP per step", ms_per_step=round(ms, 3),
                steps_per_s=round(1e3 / ms, 4), step_tflops=round(FLOP_PER_STEP_TVI2V / (ms * 1e-3) / 1e12, 1),
                frac=round(FLOP_PER_STEP_TVI2V / (ms * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS, 4), steps=steps)


def time_clip(wrapper, device, num_steps=30, scale=7.5, seed=43, tvi2v=False, fp32_vae=False):
    """One full clip: DPMPP2SAncestral (30 steps = 59 evaluations; TVI2V: 50 steps = 99) + AutoencoderKL decode -> frames/s."""
    from ccedit_amd.config import instantiate_from_config
    from ccedit_amd.sgm_compat import build_vae
    from ccedit_amd.utils.synth import fill_module_
    from ccedit_amd import ops
    vae = build_vae(device)
    fill_module_(vae, prefix="first_stage_model.")
    vae.pack(device)
    dd = "sgm.modules.diffusionmodules."
    denoiser = instantiate_from_config(dict(target=dd + "denoiser.DiscreteDenoiser", params=dict(
        num_idx=1000, weighting_config=dict(target=dd + "denoiser_weighting.EpsWeighting"),
        scaling_config=dict(target=dd + "denoiser_scaling.EpsScaling"),
        discretization_config=dict(target=dd + "discretizer.LegacyDDPMDiscretization"))))
    sampler = instantiate_from_config(dict(target=dd + "sampling.DPMPP2SAncestralSampler", params=dict(
        num_steps=num_steps, eta=1.0, s_noise=1.0, discretization_config=dict(target=dd + "discretizer.LegacyDDPMDiscretization"),
        guider_config=dict(target=dd + "guiders.VanillaCFGTV2V", params=dict(scale=scale)))))
    x, cross_c, cross_uc, hint = synth_inputs(device, seed=seed)
    c = dict(crossattn=cross_c, control_hint=hint)
    uc = dict(crossattn=cross_uc, control_hint=hint.clone())
    if tvi2v:          # the VAE-encoded reference frame, identical in c and uc (sampling_tv2v_ref.py:405-445)
        cf = (torch.randn(1, 4, H, W, generator=torch.Generator().manual_seed(seed + 100)) * 0.18215).to(device)
        c["cond_feat"], uc["cond_feat"] = cf, cf.clone()
    evals = [0]

    def network(xx, tt, cond):
        evals[0] += 1
        return wrapper(xx, tt, cond)

    wrapper.cache_hint_stem = True                 # whole-clip run: the hint stem is evaluated once per clip
    # The first stage's precision is the engine's default for the SHIPPED yamls: they set disable_first_stage_autocast, which in the
    # reference means an fp32 decode (diffusion.py:151-156), and policy vae_fp32 = 2 (the default since round 6) follows the flag.
    # `frames_per_s` is therefore the fp32-decode figure; the bf16 first stage (policy vae_fp32=0) is timed beside it as the option.
    from ccedit_amd import policy
    primary = "bf16" if policy.get("vae_fp32") == 0 else "fp32"
    other = "fp32" if primary == "bf16" else "bf16"
    # warm-up of the decoder like the W warm-up steps of the network: its first call loads code objects and grows the
    # caching allocator by ~10-20 GB (measured 0.10 s warm, 0.23-0.38 s on the first call of a fresh process)
    vae.precision = primary
    vae.decode(torch.randn(1, 4, T, H, W, device=device))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    z = sampler(lambda inp, sig, cc: denoiser(network, inp, sig, cc), x.clone(), c, uc=uc)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    zs = ops.axpby(z.contiguous(), z.contiguous(), 1.0 / 0.18215, 0.0)
    frames = vae.decode(zs)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    assert frames.shape == (1, 3, T, 8 * H, 8 * W)
    wrapper.cache_hint_stem = False
    finite = bool(torch.isfinite(frames).all())
    dec = {primary: t2 - t1}
    out2 = {}
    if fp32_vae:        # the same latent through the OTHER first stage: second decode time, distance between the two sets of frames
        vae.precision = other
        vae.decode(zs[:, :, :2].contiguous())          # code objects + allocator growth, like the warm-up above
        torch.cuda.synchronize()
        t3 = time.perf_counter()
        frames2 = vae.decode(zs)
        torch.cuda.synchronize()
        t4 = time.perf_counter()
        dec[other] = t4 - t3
        f32f = frames if primary == "fp32" else frames2
        d = (frames - frames2).double()
        out2 = dict(bf16_vs_fp32_frames_rel_rms=float(f"{float((d ** 2).mean().sqrt() / (f32f.double() ** 2).mean().sqrt()):.3g}"))
        del frames2, d
        vae.precision = primary
    if not finite:                                 # a rate for garbage is not a measurement
        sys.stderr.write("bench.py: the sampled clip contains non-finite values — frames_per_s withheld\n")
    res = dict(sampler_s=round(t1 - t0, 3), vae_decode_s=round(t2 - t1, 3), vae_precision=primary, evaluations=evals[0], sampler_steps=num_steps,
               cfg_scale=scale, hint_stem="once per clip", decoder_warmup="one untimed decode",
               vae="fp32 first stage — the reference decodes with autocast off (diffusion.py:151-156; the shipped yamls set disable_first_stage_autocast "
                   "and policy vae_fp32=2, the default, follows the flag) — its contractions as "
                   + ("six exact bf16 x bf16 products per fp32 product on v_mfma_f32_32x32x16_bf16 (exact three-way operand split, fp32 sums; policy "
                      "f32_split=1, the default: same error against fp64 and the reference's fp32 goldens as the fp32 matrix instruction)"
                      if policy.get("f32_split") else "v_mfma_f32_32x32x2_f32 (policy f32_split=0)")
                   + "; the bf16-storage first stage (policy vae_fp32=0) is the *_bf16_vae figure",
               frames_per_s=round(T / (t2 - t0), 3) if finite else None, finite=finite, **out2)
    for prec, sec in dec.items():
        res[f"vae_decode_{prec}_s"] = round(sec, 3)
        res[f"frames_per_s_{prec}_vae"] = round(T / (t1 - t0 + sec), 3) if finite else None
    if "fp32" in dec:
        res["vae_fp32_tflops"] = round(64.56 / dec["fp32"], 1)          # fp32-equivalent FLOPs of the decoder / time (the fp32 matrix peak is 157)
    return res


if __name__ == "__main__":
    main()
