"""fp32 first stage: six exact bf16 products per fp32 product (policy f32_split = 1, the default) against v_mfma_f32_32x32x2_f32.
Full-size decode (17 x 512 x 768, shipped width) of the same latent through both arms, alternating; per-shape conv timings."""
import sys, os, time
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
import torch
from ccedit_amd import hip, vae_f32 as V
from ccedit_amd.sgm_compat import build_vae
from ccedit_amd.utils.synth import fill_module_

dev = torch.device("cuda:0")
lib = hip.lib()


def arm(v):
    assert lib.ccedit_policy_set(b"f32_split", v) == 0


def timed(fn, n=3):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    return min(ts)


print("== 3x3 convolutions, one frame-batch each (us, TFLOP/s fp32-equivalent) ==")
for (frames, h, w, cin, cout, up) in [(17, 512, 768, 128, 128, False), (17, 256, 384, 256, 256, False), (17, 256, 384, 256, 256, True),
                                      (17, 128, 192, 512, 512, False), (17, 128, 192, 512, 512, True), (17, 64, 96, 512, 512, False),
                                      (17, 512, 768, 256, 128, False), (17, 512, 768, 128, 3, False)]:
    g = torch.Generator().manual_seed(1)
    x = torch.randn(frames, h, w, cin, generator=g).to(dev)
    pw = V.pack_f32(torch.randn(cout, cin, 3, 3, generator=g) * (9 * cin) ** -0.5, torch.randn(cout, generator=g), dev)
    ho, wo = (2 * h, 2 * w) if up else (h, w)
    fl = 2.0 * frames * ho * wo * cin * cout * 9
    row = []
    outs = {}
    for v in (1, 0, 1, 0):
        arm(v)
        t = timed(lambda: outs.__setitem__(v, V.conv2d_f32(x, pw, upsample=up)))
        row.append((v, t))
    d = (outs[1].double() - outs[0].double())
    rel = float((d ** 2).mean().sqrt() / (outs[0].double() ** 2).mean().sqrt())
    del outs, x
    torch.cuda.empty_cache()
    print(f"{cin:4d}->{cout:4d} {h}x{w}{' up' if up else '   '}: " + "  ".join(f"{'split' if v else 'mfma32'} {t * 1e6:8.0f} us {fl / t / 1e12:6.1f}" for v, t in row)
          + f"   arms apart {rel:.2e}", flush=True)

vae = build_vae(dev)
fill_module_(vae, prefix="first_stage_model.")
vae.pack(dev)
vae.precision = "fp32"
T = int(sys.argv[1]) if len(sys.argv) > 1 else 17
z = torch.randn(1, 4, T, 64, 96, device=dev)
out = {}
print("== full-size decode ==")
for v in (1, 0, 1, 0):
    arm(v)
    vae.decode(z[:, :, :2].contiguous())
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out[v] = vae.decode(z)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"{'six bf16 products' if v else 'v_mfma_f32_32x32x2_f32'}: decode of {T} frames {dt * 1e3:.1f} ms = {64.56 * T / 17 / dt:.1f} TFLOP/s (fp32-equivalent)", flush=True)
d = (out[1] - out[0]).double()
print(f"the two arms' frames: rel rms {float((d ** 2).mean().sqrt() / (out[0].double() ** 2).mean().sqrt()):.3e}, finite {bool(torch.isfinite(out[1]).all())}")
arm(1)
