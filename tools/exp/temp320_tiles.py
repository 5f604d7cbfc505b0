#!/usr/bin/env python3
"""Conv1d k3 over T at 320 channels (64x96 level, 2 clips x 17 frames): block shapes of ccedit_gemm, graph-replayed.
   python tools/exp/temp320_tiles.py"""
import os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
import torch
from ccedit_amd import ops, hip
from ccedit_amd.packing import pack_weight
BF = torch.bfloat16


def timeit(f, n=10):
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


B, T, H, W, C = 2, 17, 64, 96, 320
x = torch.randn(B * T, H, W, C, device="cuda").to(BF)
r = torch.randn(B * T * H * W, C, device="cuda").to(BF)
pw = pack_weight(torch.randn(C, C, 3) * (3 * C) ** -0.5, torch.randn(C)).to("cuda")
fl = 2.0 * B * T * H * W * C * C * 3
for tile in (14, 0, 1, 2, 3, 4, 5, 6, 11, 12, 13):
    try:
        t = timeit(lambda: ops.conv_temporal(x, T, pw, res1=r, tile=tile))
        print(f"tile {tile:2d}: {t:7.1f} us  {fl / t / 1e6:6.0f} TF/s   [{hip.lib().ccedit_last_kernel().decode()}]", flush=True)
    except Exception as e:
        print(f"tile {tile:2d}: {str(e)[:100]}", flush=True)
