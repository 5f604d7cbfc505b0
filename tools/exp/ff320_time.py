"""Fused dim-320 feed-forward vs LayerNorm + two GEMMs at the 64x96 level (34 frames): us per call, TF/s.
Inputs rotate over several buffers so that x arrives cold (as in the network)."""
import sys, os
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
import torch
from ccedit_amd import ops
from ccedit_amd.packing import pack_ff320, pack_weight

m = int(sys.argv[1]) if len(sys.argv) > 1 else 34 * 6144
g = torch.Generator().manual_seed(0)
w1, b1 = torch.randn(2560, 320, generator=g) * 320 ** -0.5, torch.randn(2560, generator=g) * 0.1
w2, b2 = torch.randn(320, 1280, generator=g) * 1280 ** -0.5, torch.randn(320, generator=g) * 0.1
lg, lb = 1 + 0.1 * torch.randn(320, generator=g), 0.1 * torch.randn(320, generator=g)
pk = pack_ff320(w1, b1, w2, b2, lg, lb, device="cuda")
p1, p2 = pack_weight(w1, b1, geglu=True).to("cuda"), pack_weight(w2, b2).to("cuda")
lgc, lbc = lg.cuda(), lb.cuda()
xs = [torch.randn(m, 320, device="cuda").to(torch.bfloat16) for _ in range(4)]
flops = m * (2.0 * 320 * 2560 + 2.0 * 1280 * 320)


def fused(x):
    return ops.ff320(x, pk)


def unfused(x):
    n = ops.layernorm(x, lgc, lbc, 1e-5)
    h = ops.linear(n, p1)
    return ops.linear(h, p2, res1=x)


for name, fn in (("fused", fused), ("unfused", unfused), ("fused", fused), ("unfused", unfused)):
    for i in range(3):
        fn(xs[i % 4])
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 12
    e0.record()
    for i in range(reps):
        fn(xs[i % 4])
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    print(f"{name:8s} m={m}: {us:9.1f} us  {flops / us / 1e6:8.1f} TF/s")
a, b = fused(xs[0]).float(), unfused(xs[0]).float()
print("branch rel rms fused vs unfused:", ((a - b).pow(2).mean().sqrt() / (b - xs[0].float()).pow(2).mean().sqrt()).item())
