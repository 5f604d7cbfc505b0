#!/usr/bin/env python3
"""lin640s_kernel: what a launch costs against the number of channel slices (N = 256 .. 768 at M = 34 x 1536), operands rotating
over NB buffers (cold, as in the network).  The question: is the N = 640 launch (3 slices, the third half empty, 30 of 32 workgroups
per XCD) bound by the slices' re-fetch of the activations / per-tile latency, or by matrix work?"""
import os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
import torch
from ccedit_amd import ops, hip
from ccedit_amd.packing import pack_weight
BF = torch.bfloat16
M = 34 * 1536
NB = 6


def timeit(f, n=24):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for i in range(NB):
            f(i)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for i in range(n):
                f(i)
        g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


xs = [torch.randn(M, 640, device="cuda").to(BF) for _ in range(NB)]
for tile in (10, 0):
    for n in (128, 256, 384, 512, 640, 768, 1280, 1920):
        rs = [torch.randn(M, n, device="cuda").to(BF) for _ in range(NB)]
        pw = pack_weight(torch.randn(n, 640) * 0.04, torch.randn(n)).to("cuda")
        outs = [torch.empty(M, n, dtype=BF, device="cuda") for _ in range(NB)]
        fl = 2 * M * 640 * n
        try:
            t = timeit(lambda i: ops.linear(xs[i % NB], pw, out=outs[i % NB], tile=tile))
            k = hip.lib().ccedit_last_kernel().decode()
            t2 = timeit(lambda i: ops.linear(xs[i % NB], pw, res1=rs[i % NB], out=outs[i % NB], tile=tile))
            by = M * (640 + n) * 2
            print(f"tile {tile:2d} N={n:4d}: plain {t:7.1f} us {fl / t / 1e6:6.0f} TF/s {by / t / 1e6:5.2f} TB/s | res {t2:7.1f} us {fl / t2 / 1e6:6.0f} TF/s {(by + M * n * 2) / t2 / 1e6:5.2f} TB/s  [{k}]", flush=True)
        except Exception as e:
            print(f"tile {tile} N={n}: {e}")
        del rs, outs
