"""Full-size first-stage ENCODE (17 x 512 x 768 frames -> 64 x 96 latents) in fp32: six bf16 products against the fp32 matrix instruction"""
import sys, os, time
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
import torch
from ccedit_amd import hip
from ccedit_amd.sgm_compat import build_vae
from ccedit_amd.utils.synth import fill_module_
dev = torch.device("cuda:0")
vae = build_vae(dev)
fill_module_(vae, prefix="first_stage_model.")
vae.pack(dev)
vae.precision = "fp32"
T = int(sys.argv[1]) if len(sys.argv) > 1 else 17
x = (torch.rand(1, 3, T, 512, 768, generator=torch.Generator().manual_seed(5)) * 2 - 1).to(dev)
noise = torch.randn(T, 4, 64, 96, generator=torch.Generator().manual_seed(6))          # (B*T, z, H/8, W/8)
out = {}
for split in (1, 0):
    assert hip.lib().ccedit_policy_set(b"f32_split", split) == 0
    vae.encode(x, noise=noise)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out[split] = vae.encode(x, noise=noise)
    torch.cuda.synchronize()
    print(f"{'six bf16 products' if split else 'v_mfma_f32_32x32x2_f32'}: encode of {T} frames {(time.perf_counter() - t0) * 1e3:.1f} ms, finite {bool(torch.isfinite(out[split]).all())}", flush=True)
hip.lib().ccedit_policy_set(b"f32_split", 1)
d = (out[1] - out[0]).double()
print(f"latents of the two arms: rel rms {float((d ** 2).mean().sqrt() / (out[0].double() ** 2).mean().sqrt()):.3e}")
