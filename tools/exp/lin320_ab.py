#!/usr/bin/env python3
"""K = 320 Linears of the 64x96 level, timed in isolation (run twice: CCEDIT_LIN320S=0 / 1 for the A/B).
   python tools/exp/lin320_ab.py"""
import os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
import torch
from ccedit_amd import ops
from ccedit_amd.packing import pack_weight, fold_layernorm
BF = torch.bfloat16
M = 34 * 6144


def timeit(f, n=20):
    """n launches replayed from a HIP graph (host overhead out of the picture)"""
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3):
            f()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(n):
                f()
        g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


x = torch.randn(M, 320, device="cuda").to(BF)
flush = torch.empty(512 << 20, dtype=torch.uint8, device="cuda")
for n in (320, 640, 960):
    r = torch.randn(M, n, device="cuda").to(BF)
    pw = pack_weight(torch.randn(n, 320) * 0.05, torch.randn(n)).to("cuda")
    out = torch.empty(M, n, dtype=BF, device="cuda")
    mb = (M * 320 * 2 + M * n * 2) / 1e6
    t = timeit(lambda: ops.linear(x, pw, out=out))
    print(f"plain   N={n:4d}: {t:7.1f} us  {mb / t:6.2f} TB/s  {2 * M * 320 * n / t / 1e6:6.0f} TF/s")
    t = timeit(lambda: ops.linear(x, pw, res1=r, out=out))
    print(f"res     N={n:4d}: {t:7.1f} us  {(mb + M * n * 2 / 1e6) / t:6.2f} TB/s")
    pl = fold_layernorm([torch.randn(n, 320) * 0.05], None, torch.ones(320), torch.zeros(320), device="cuda")
    if not ops.ln320_applicable(M, pl):
        continue
    t = timeit(lambda: ops.linear(x, pl, ln_eps=1e-5, out=out))
    print(f"ln      N={n:4d}: {t:7.1f} us  {mb / t:6.2f} TB/s")
print("kernels:", os.environ.get("CCEDIT_LIN320S", "1"))
