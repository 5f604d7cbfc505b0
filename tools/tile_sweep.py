#!/usr/bin/env python3
"""Sweep the GEMM block shapes over the hot path's main contraction shapes (TF/s per tile id)."""
import sys, os
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch
from ccedit_amd import ops
from ccedit_amd.packing import pack_weight
from tools.microbench import timeit, rnd
dev = "cuda"
N = 34
rows = []

def conv(name, h, w, cin, cout, stride=1, up=False):
    x = rnd(N, h, w, cin)
    pw = pack_weight(torch.randn(cout, cin, 3, 3) * (9 * cin) ** -0.5, torch.randn(cout)).to(dev)
    ho, wo = (2 * h, 2 * w) if up else (h // stride, w // stride)
    fl = 2.0 * N * ho * wo * cout * cin * 9
    r = []
    for tile in (1, 2, 3, 4, 5):
        t = timeit(lambda: ops.conv2d(x, pw, stride=stride, upsample=up, tile=tile), iters=5, warm=2)
        r.append(fl / t / 1e12)
    print(f"{name:34s} M={N*ho*wo:7d} K={9*cin:6d} N={cout:5d}  " + "  ".join(f"t{i+1}:{v:6.0f}" for i, v in enumerate(r)), flush=True)

def lin(name, m, k, n, geglu=False):
    x = rnd(m, k)
    pw = pack_weight(torch.randn(n, k) * k ** -0.5, torch.randn(n), geglu=geglu).to(dev)
    fl = 2.0 * m * k * n
    r = []
    for tile in (1, 2, 3, 4, 5):
        t = timeit(lambda: ops.linear(x, pw, tile=tile), iters=5, warm=2)
        r.append(fl / t / 1e12)
    print(f"{name:34s} M={m:7d} K={k:6d} N={n:5d}  " + "  ".join(f"t{i+1}:{v:6.0f}" for i, v in enumerate(r)), flush=True)

def temp(name, h, w, c):
    x = rnd(N, h, w, c)
    pw = pack_weight(torch.randn(c, c, 3) * (3 * c) ** -0.5, torch.randn(c)).to(dev)
    fl = 2.0 * N * h * w * c * c * 3
    r = []
    for tile in (1, 2, 3, 4, 5):
        t = timeit(lambda: ops.conv_temporal(x, 17, pw, tile=tile), iters=5, warm=2)
        r.append(fl / t / 1e12)
    print(f"{name:34s} M={N*h*w:7d} K={3*c:6d} N={c:5d}  " + "  ".join(f"t{i+1}:{v:6.0f}" for i, v in enumerate(r)), flush=True)

conv("conv L0 320->320", 64, 96, 320, 320)
conv("conv L0 640->320 (dec)", 64, 96, 640, 320)
conv("conv L0 960->320 (dec)", 64, 96, 960, 320)
conv("conv L0 up 640->640", 32, 48, 640, 640, up=True)
conv("conv L1 640->640", 32, 48, 640, 640)
conv("conv L1 320->640", 32, 48, 320, 640)
conv("conv L1 1280->640 (dec)", 32, 48, 1280, 640)
conv("conv L1 1920->640 (dec)", 32, 48, 1920, 640)
conv("conv L1 up 1280->1280", 16, 24, 1280, 1280, up=True)
conv("conv L2 1280->1280", 16, 24, 1280, 1280)
conv("conv L2 2560->1280 (dec)", 16, 24, 2560, 1280)
conv("conv L3 1280->1280", 8, 12, 1280, 1280)
conv("conv L3 2560->1280", 8, 12, 2560, 1280)
conv("down L0", 64, 96, 320, 320, stride=2)
M0, M1, M2, M3 = N * 6144, N * 1536, N * 384, N * 96
lin("L0 proj 320->320", M0, 320, 320)
lin("L0 qkv 320->960", M0, 320, 960)
lin("L0 kv 320->640", M0, 320, 640)
lin("L0 ff.proj 320->2560 geglu", M0, 320, 2560, True)
lin("L0 ff.out 1280->320", M0, 1280, 320)
lin("L0 skip 1x1 960->320", M0, 960, 320)
lin("L1 proj 640->640", M1, 640, 640)
lin("L1 qkv 640->1920", M1, 640, 1920)
lin("L1 ff.proj 640->5120 geglu", M1, 640, 5120, True)
lin("L1 ff.out 2560->640", M1, 2560, 640)
lin("L2 proj 1280->1280", M2, 1280, 1280)
lin("L2 qkv 1280->3840", M2, 1280, 3840)
lin("L2 ff.proj 1280->10240 geglu", M2, 1280, 10240, True)
lin("L2 ff.out 5120->1280", M2, 5120, 1280)
lin("L3 qkv 1280->3840", M3, 1280, 3840)
lin("L3 ff.proj geglu", M3, 1280, 10240, True)
temp("temporal L0 320", 64, 96, 320)
temp("temporal L1 640", 32, 48, 640)
temp("temporal L2 1280", 16, 24, 1280)
temp("temporal L3 1280", 8, 12, 1280)
