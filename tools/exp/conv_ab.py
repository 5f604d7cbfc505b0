#!/usr/bin/env python3
"""3x3 convs of the 64x96 / 32x48 levels, timed in isolation from a HIP graph: tile 8 (LDS-halo kernel) vs tile 1 (tap gather) vs the
   automatic choice.  (The 16 x 16-rectangle halo variant of DESIGN.md section 3.3 was measured with this script and removed.)
   python tools/exp/conv_ab.py"""
import os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
import torch
from ccedit_amd import ops, hip
from ccedit_amd.packing import pack_weight
BF = torch.bfloat16
N = 34


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


for (h, w, cin, cout) in ((64, 96, 320, 320), (64, 96, 640, 320), (64, 96, 960, 320), (32, 48, 640, 640), (32, 48, 320, 640), (32, 48, 1280, 640), (32, 48, 1920, 640)):
    x = torch.randn(N, h, w, cin, device="cuda").to(BF)
    pw = pack_weight(torch.randn(cout, cin, 3, 3) * (9 * cin) ** -0.5, torch.randn(cout)).to("cuda")
    fl = 2.0 * N * h * w * cout * cin * 9
    row = f"{h}x{w} {cin:4d}->{cout:4d}:"
    for tile in (8, 1, 0):
        t = timeit(lambda: ops.conv2d(x, pw, tile=tile, gn=True))
        row += f"  tile {tile}: {t:7.1f} us {fl / t / 1e6:6.0f} TF/s"
    print(row + f"   [{hip.lib().ccedit_last_kernel().decode()}]", flush=True)
