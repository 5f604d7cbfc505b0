"""Block tail (to_out + FF + proj_out in ONE ff320 launch) vs the three launches it replaces at the 64x96 level (34 frames):
us per call.  Inputs rotate over several buffers so that they arrive cold (as in the network)."""
import sys, os
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
import torch
from ccedit_amd import ops
from ccedit_amd.packing import pack_ff320, pack_ff320_tail, pack_weight

m = int(sys.argv[1]) if len(sys.argv) > 1 else 34 * 6144
g = torch.Generator().manual_seed(0)
w1, b1 = torch.randn(2560, 320, generator=g) * 320 ** -0.5, torch.randn(2560, generator=g) * 0.1
w2, b2 = torch.randn(320, 1280, generator=g) * 1280 ** -0.5, torch.randn(320, generator=g) * 0.1
wo, bo = torch.randn(320, 320, generator=g) * 320 ** -0.5, torch.randn(320, generator=g) * 0.1
wp, bp = torch.randn(320, 320, generator=g) * 320 ** -0.5, torch.randn(320, generator=g) * 0.1
lg, lb = 1 + 0.1 * torch.randn(320, generator=g), 0.1 * torch.randn(320, generator=g)
base = pack_ff320(w1, b1, w2, b2, lg, lb, device="cuda")
full = pack_ff320_tail(base, wo, bo, wp, bp, device="cuda")
pro = pack_ff320_tail(base, wo, bo, device="cuda")
po, pp = pack_weight(wo, bo).to("cuda"), pack_weight(wp, bp).to("cuda")
mk = lambda: [torch.randn(m, 320, device="cuda").to(torch.bfloat16) for _ in range(3)]
a_s, r_s, x_s = mk(), mk(), mk()


def three(i):
    tok = ops.linear(a_s[i], po, res1=r_s[i])
    return ops.linear(ops.ff320(tok, base), pp, res1=x_s[i])


def two(i):
    return ops.linear(ops.ff320(None, pro, a=a_s[i], res=r_s[i]), pp, res1=x_s[i])


def one(i):
    return ops.ff320(None, full, a=a_s[i], res=r_s[i], res2=x_s[i])


def ff_only(i):
    return ops.ff320(a_s[i], base)


for name, fn in (("three launches", three), ("to_out fused", two), ("block tail", one), ("ff320 alone", ff_only)) * 2:
    for i in range(3):
        fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 12
    e0.record()
    for i in range(reps):
        fn(i % 3)
    e1.record()
    torch.cuda.synchronize()
    print(f"{name:16s} m={m}: {e0.elapsed_time(e1) * 1e3 / reps:9.1f} us")
y1, y3 = one(0).float(), three(0).float()
print("rel rms block tail vs three launches:", ((y1 - y3).pow(2).mean().sqrt() / y3.pow(2).mean().sqrt()).item())
