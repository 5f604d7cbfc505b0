"""Full-size first-stage decode (17 x 512 x 768): the bf16-storage first stage, the fp32 first stage as the product runs it (six exact
bf16 products per fp32 product, policy f32_split = 1) and on the fp32 matrix instruction (f32_split = 0); wall time and memory."""
import sys, os, time
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch
from ccedit_amd import hip
from ccedit_amd.sgm_compat import build_vae
from ccedit_amd.utils.synth import fill_module_

dev = torch.device("cuda:0")
vae = build_vae(dev)
fill_module_(vae, prefix="first_stage_model.")
vae.pack(dev)
T = int(sys.argv[1]) if len(sys.argv) > 1 else 17
z = torch.randn(1, 4, T, 64, 96, device=dev)
out = {}
for name, prec, split in (("bf16", "bf16", 1), ("fp32 (six bf16 products)", "fp32", 1), ("fp32 (v_mfma_f32_32x32x2_f32)", "fp32", 0)):
    vae.precision = prec
    assert hip.lib().ccedit_policy_set(b"f32_split", split) == 0
    vae.decode(z)                      # (one untimed decode of the same size, as bench.py's clip: the caching allocator holds the blocks afterwards)
    torch.cuda.synchronize()
    torch.cuda.reset_peak_memory_stats()
    t0 = time.perf_counter()
    out[name] = vae.decode(z)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"{name}: decode of {T} frames {dt * 1e3:.1f} ms = {64.56 * T / 17 / dt:.1f} TFLOP/s{'' if prec == 'bf16' else ' fp32-equivalent'}, "
          f"peak memory {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB", flush=True)
hip.lib().ccedit_policy_set(b"f32_split", 1)
ref = out["fp32 (v_mfma_f32_32x32x2_f32)"].double()
for name in ("bf16", "fp32 (six bf16 products)"):
    d = out[name].double() - ref
    print(f"{name} vs fp32 (v_mfma_f32_32x32x2_f32) frames: rel rms {float((d ** 2).mean().sqrt() / (ref ** 2).mean().sqrt()):.3e}, finite {bool(torch.isfinite(out[name]).all())}")
