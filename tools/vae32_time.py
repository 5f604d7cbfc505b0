"""Full-size first-stage decode (17 x 512 x 768): bf16 default against the fp32 option (policy vae_fp32), wall time and memory."""
import sys, os, time
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch
from ccedit_amd.sgm_compat import build_vae
from ccedit_amd.utils.synth import fill_module_

dev = torch.device("cuda:0")
vae = build_vae(dev)
fill_module_(vae, prefix="first_stage_model.")
vae.pack(dev)
T = int(sys.argv[1]) if len(sys.argv) > 1 else 17
z = torch.randn(1, 4, T, 64, 96, device=dev)
out = {}
for prec in ("bf16", "fp32"):
    vae.precision = prec
    vae.decode(z[:, :, :2].contiguous())
    torch.cuda.synchronize()
    torch.cuda.reset_peak_memory_stats()
    t0 = time.perf_counter()
    out[prec] = vae.decode(z)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"{prec}: decode of {T} frames {dt * 1e3:.1f} ms = {64.56 * T / 17 / dt:.1f} TFLOP/s, peak memory {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB", flush=True)
d = (out["bf16"] - out["fp32"]).double()
print(f"bf16 vs fp32 frames: rel rms {float((d ** 2).mean().sqrt() / (out['fp32'].double() ** 2).mean().sqrt()):.3e}, finite {bool(torch.isfinite(out['fp32']).all())}")
