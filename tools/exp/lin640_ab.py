#!/usr/bin/env python3
"""K = 640 Linears of the 32x48 level, timed in isolation (run twice: CCEDIT_LIN640=0 / 1 for the A/B).
   python tools/exp/lin640_ab.py"""
import os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
import torch
from ccedit_amd import ops, hip
from ccedit_amd.packing import pack_weight, fold_layernorm
BF = torch.bfloat16
M = 34 * 1536
TILE = int(os.environ.get("TILE", "0"))       # 10 forces lin640s_kernel (automatic only from 1024 output channels), 12 / 13 gemm8p


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


x = torch.randn(M, 640, device="cuda").to(BF)
st = ops.row_stats(x, 1e-5)
for n in (640, 1280, 1920):
    r = torch.randn(M, n, device="cuda").to(BF)
    pw = pack_weight(torch.randn(n, 640) * 0.04, torch.randn(n)).to("cuda")
    out = torch.empty(M, n, dtype=BF, device="cuda")
    fl = 2 * M * 640 * n
    t = timeit(lambda: ops.linear(x, pw, out=out, tile=TILE))
    print(f"plain     N={n:4d}: {t:7.1f} us  {fl / t / 1e6:6.0f} TF/s   [{hip.lib().ccedit_last_kernel().decode()}]")
    t = timeit(lambda: ops.linear(x, pw, res1=r, out=out, tile=TILE))
    print(f"res       N={n:4d}: {t:7.1f} us  {fl / t / 1e6:6.0f} TF/s")
    t = timeit(lambda: ops.linear(x, pw, res1=r, out=out, row_sums=True, tile=TILE))
    print(f"res+sums  N={n:4d}: {t:7.1f} us  {fl / t / 1e6:6.0f} TF/s")
    pl = fold_layernorm([torch.randn(n, 640) * 0.04], [torch.randn(n)], torch.ones(640), torch.zeros(640)).to("cuda")
    t = timeit(lambda: ops.linear(x, pl, ln_stats=st, out=out, tile=TILE))
    print(f"ln        N={n:4d}: {t:7.1f} us  {fl / t / 1e6:6.0f} TF/s   [{hip.lib().ccedit_last_kernel().decode()}]")
print("CCEDIT_LIN640 =", os.environ.get("CCEDIT_LIN640", "1"))
