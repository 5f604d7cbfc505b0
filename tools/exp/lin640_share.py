#!/usr/bin/env python3
"""lin640s at 1, 2, 5 and 15 channel slices: time per tile and the traffic it implies if every slice fetched its own copy of the
activations.  (The experiments of DESIGN.md section 3.3 — start skew, leader / follower slices, re-reading L2-resident tiles — ran
this script against builds with temporary switches that have been removed.)"""
import os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
import torch
from ccedit_amd import ops, hip
from ccedit_amd.packing import pack_weight
BF = torch.bfloat16
M = int(sys.argv[1]) if len(sys.argv) > 1 else 34 * 1536


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
for n in (128, 256, 640, 1920):
    pw = pack_weight(torch.randn(n, 640) * 0.04, torch.randn(n)).to("cuda")
    out = torch.empty(M, n, dtype=BF, device="cuda")
    t = timeit(lambda: ops.linear(x, pw, out=out, tile=10))
    tiles = M / 32 / 8 / (32 // (n // 128))
    print(f"flags={os.environ.get('CCEDIT_L640_FLAGS', '0'):>4s} M={M} N={n:4d}: {t:8.1f} us  {t / tiles:6.3f} us/tile  {2 * M * 640 * n / t / 1e6:6.0f} TF/s"
          f"  x once + out: {(M * 640 * 2 + M * n * 2) / t / 1e6:5.2f} TB/s, x per slice: {(M * 640 * 2 * (n // 128) + M * n * 2) / t / 1e6:5.2f} TB/s")
