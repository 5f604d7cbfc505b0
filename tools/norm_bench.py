#!/usr/bin/env python3
"""Time the normalisation kernels at the four latent levels of the TV2V workload (B=2, T=17)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ccedit_amd import ops

def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    # rotate through enough buffers that the Infinity Cache does not hold the operands
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

B, T = 2, 17
for (h, w, c) in [(64, 96, 320), (32, 48, 640), (16, 24, 1280), (8, 12, 1280)]:
    xs = [torch.randn(B * T, h, w, c, device="cuda").to(torch.bfloat16) for _ in range(6 if h >= 32 else 12)]
    g, b = torch.ones(c, device="cuda"), torch.zeros(c, device="cuda")
    mb = xs[0].numel() * 2 / 1e6
    i = [0]
    def nxt():
        i[0] = (i[0] + 1) % len(xs); return xs[i[0]]
    ts = timeit(lambda: ops.groupnorm_spatial(nxt(), g, b, 1e-5, True))
    tt = timeit(lambda: ops.groupnorm_temporal(nxt(), B, T, g, b, 1e-5, True))
    tl = timeit(lambda: ops.layernorm(nxt().view(-1, c), g, b))
    print(f"{h}x{w} C={c} ({mb:.0f} MB): spatial GN {ts:.1f} us ({3*mb/ts/1e3:.2f} TB/s of 3 passes)  "
          f"temporal GN {tt:.1f} us ({2*mb/tt/1e3:.2f} TB/s of 2 passes)  LN {tl:.1f} us ({2*mb/tl/1e3:.2f} TB/s)")

# concatenation + statistics (decoder skip joins) at the shapes of the step
for (hw_, c1, c2) in [((64, 96), 320, 320), ((64, 96), 640, 320), ((32, 48), 1280, 640), ((32, 48), 640, 640), ((16, 24), 1280, 1280), ((8, 12), 1280, 1280)]:
    h, w = hw_
    n = B * T
    sets = [(torch.randn(n, h, w, c1, device="cuda").to(torch.bfloat16), torch.randn(n, h, w, c2, device="cuda").to(torch.bfloat16),
             torch.randn(n, h, w, c2, device="cuda").to(torch.bfloat16)) for _ in range(4 if h >= 32 else 10)]
    i = [0]
    def nxt3():
        i[0] = (i[0] + 1) % len(sets); return sets[i[0]]
    t = timeit(lambda: ops.cat_add(*nxt3(), gn=True))
    mb = n * h * w * (2 * c1 + 3 * c2) * 2 / 1e6
    print(f"cat_add_gn {h}x{w} {c1}+{c2} ({mb:.0f} MB): {t:.1f} us ({mb/t/1e3:.2f} TB/s)")
