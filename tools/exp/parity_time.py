#!/usr/bin/env python3
"""The four parity convs of upsample + conv 3x3 at the network's three up-sampling shapes: persistent gather kernel (automatic) against
the tap-gather kernel (tile 1), us per upsample (four launches), cold operands."""
import os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
import torch
from ccedit_amd import ops, hip
from ccedit_amd.packing import pack_upsample_parities
BF = torch.bfloat16
NB = 4


def timeit(f, n=8):
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


for n, h, w, c in ((34, 8, 12, 1280), (34, 16, 24, 1280), (34, 32, 48, 640)):
    xs = [torch.randn(n, h, w, c, device="cuda").to(BF) for _ in range(NB)]
    par = [pack_upsample_parities(torch.randn(c, c, 3, 3) * (9 * c) ** -0.5, torch.randn(c), device="cuda") for _ in range(NB)]
    fl = 2.0 * n * h * w * 4 * c * c * 4
    row = f"{n} x {h} x {w} x {c}:"
    for tile in (0, 1, 0, 1):
        t = timeit(lambda i: ops.conv2d_upsampled(xs[i % NB], par[i % NB], tile=tile))
        k = hip.lib().ccedit_last_kernel().decode()
        row += f"  tile {tile}: {t:7.1f} us {fl / t / 1e6:5.0f} TF/s executed"
        if tile == 0:
            row += f" [{k[:48]}]"
    print(row, flush=True)
