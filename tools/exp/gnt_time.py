"""temporal GroupNorm: us per launch with / without SiLU, against LayerNorm and a plain copy of the same bytes."""
import sys, os
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
import torch
from ccedit_amd import ops


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


B, T = 2, 17
for (h, w, c) in [(64, 96, 320), (32, 48, 640), (16, 24, 1280)]:
    xs = [torch.randn(B * T, h, w, c, device="cuda").to(torch.bfloat16) for _ in range(6)]
    g, b = torch.ones(c, device="cuda"), torch.zeros(c, device="cuda")
    i = [0]

    def nxt():
        i[0] = (i[0] + 1) % len(xs)
        return xs[i[0]]
    mb = xs[0].numel() * 2 / 1e6
    r = {}
    r["gn_t silu"] = timeit(lambda: ops.groupnorm_temporal(nxt(), B, T, g, b, 1e-5, True))
    r["gn_t plain"] = timeit(lambda: ops.groupnorm_temporal(nxt(), B, T, g, b, 1e-5, False))
    r["layernorm"] = timeit(lambda: ops.layernorm(nxt().view(-1, c), g, b))
    r["silu pass"] = timeit(lambda: ops.silu(nxt()))
    r["torch copy"] = timeit(lambda: nxt().clone())
    print(f"{h}x{w} C={c} ({mb:.0f} MB): " + "  ".join(f"{k} {v:.1f}us={2 * mb / v / 1e3:.2f}TB/s" for k, v in r.items()), flush=True)
