#!/usr/bin/env python3
"""GEMM block-shape sweep under in-network conditions: every call reads an activation that a producer kernel
(LayerNorm / GroupNorm-apply sized pass) has just written, rotating through enough buffer sets that nothing but
the producer's write can still be in the 256 MB Infinity Cache — the isolated sweep (tools/tile_sweep.py) re-reads
hot operands and over-rates the wide tiles.  Prints TF/s per tile id."""
import sys, os
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch
from ccedit_amd import ops
from ccedit_amd.packing import pack_weight
dev = "cuda"
TILES = [int(t) for t in os.environ.get("TILES", "1,2,3,4,5").split(",")]
NB = 6


def bench(name, make_call, flops, m, k):
    """make_call(i, tile) -> (producer thunk, gemm thunk) on buffer set i."""
    res = []
    for tile in TILES:
        try:
            calls = [make_call(i, tile) for i in range(NB)]
            for p, g in calls[:2]:
                p(); g()
            torch.cuda.synchronize()
            tot = 0.0
            evs = []
            for rep in range(2):
                for p, g in calls:
                    p()
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(); g(); e1.record()
                    evs.append((e0, e1))
            torch.cuda.synchronize()
            t = sum(a.elapsed_time(b) for a, b in evs) / len(evs) * 1e-3
            res.append(flops / t / 1e12)
        except Exception as e:
            print("   ", type(e).__name__, str(e)[:200], file=sys.stderr)
            res.append(float("nan"))
    print(f"{name:40s} " + "  ".join(f"t{t}:{v:6.0f}" for t, v in zip(TILES, res)), flush=True)


def lin(name, m, k, n, geglu=False, res=False):
    pw = pack_weight(torch.randn(n, k) * k ** -0.5, torch.randn(n), geglu=geglu).to(dev)
    src = [torch.randn(m, k, device=dev).to(torch.bfloat16) for _ in range(NB)]
    a = [torch.empty_like(s) for s in src]
    r = [torch.randn(m, n, device=dev).to(torch.bfloat16) for _ in range(NB)] if res else [None] * NB
    g, b = torch.ones(k, device=dev), torch.zeros(k, device=dev)

    def mk(i, tile):
        def prod():
            hip_ln(src[i], a[i], g, b)
        return prod, (lambda: ops.linear(a[i], pw, res1=r[i], tile=tile))
    bench(name, mk, 2.0 * m * k * n, m, k)


def hip_ln(x, y, g, b):
    from ccedit_amd import hip
    hip.check(hip.lib().ccedit_layernorm(x.data_ptr(), y.data_ptr(), g.data_ptr(), b.data_ptr(), x.shape[0], x.shape[1], 1e-5,
                                         torch.cuda.current_stream().cuda_stream), "ln")


def conv(name, n, h, w, cin, cout):
    pw = pack_weight(torch.randn(cout, cin, 3, 3) * (9 * cin) ** -0.5, torch.randn(cout)).to(dev)
    src = [torch.randn(n * h * w, cin, device=dev).to(torch.bfloat16) for _ in range(NB)]
    a = [torch.empty_like(s) for s in src]
    g, b = torch.ones(cin, device=dev), torch.zeros(cin, device=dev)

    def mk(i, tile):
        return (lambda: hip_ln(src[i], a[i], g, b)), (lambda: ops.conv2d(a[i].view(n, h, w, cin), pw, tile=tile))
    bench(name, mk, 2.0 * n * h * w * cout * cin * 9, n * h * w, 9 * cin)


def temp(name, n, h, w, c):
    pw = pack_weight(torch.randn(c, c, 3) * (3 * c) ** -0.5, torch.randn(c)).to(dev)
    src = [torch.randn(n * h * w, c, device=dev).to(torch.bfloat16) for _ in range(NB)]
    a = [torch.empty_like(s) for s in src]
    g, b = torch.ones(c, device=dev), torch.zeros(c, device=dev)

    def mk(i, tile):
        return (lambda: hip_ln(src[i], a[i], g, b)), (lambda: ops.conv_temporal(a[i].view(n, h, w, c), 17, pw, res1=src[i], tile=tile))
    bench(name, mk, 2.0 * n * h * w * c * c * 3, n * h * w, 3 * c)


M0, M1, M2 = 34 * 6144, 34 * 1536, 34 * 384
sel = sys.argv[1:] or ["all"]
def want(k): return "all" in sel or k in sel
if want("l0"):
    lin("L0 to_out 320->320 +res", M0, 320, 320, res=True)
    lin("L0 proj 320->320", M0, 320, 320)
    lin("L0 qkv 320->960", M0, 320, 960)
    lin("L0 GEGLU 320->2560", M0, 320, 2560, geglu=True)
    lin("L0 ff.out 1280->320 +res", M0, 1280, 320, res=True)
if want("l1"):
    lin("L1 to_out 640->640 +res", M1, 640, 640, res=True)
    lin("L1 GEGLU 640->5120", M1, 640, 5120, geglu=True)
    lin("L1 ff.out 2560->640 +res", M1, 2560, 640, res=True)
if want("l2"):
    lin("L2 to_out 1280->1280 +res", M2, 1280, 1280, res=True)
    lin("L2 GEGLU 1280->10240", M2, 1280, 10240, geglu=True)
    lin("L2 ff.out 5120->1280 +res", M2, 5120, 1280, res=True)
    lin("L2 qkv 1280->3840", M2, 1280, 3840)
    lin("L2 proj 1280->1280", M2, 1280, 1280)
if want("l3"):
    M3 = 34 * 96
    lin("L3 to_out 1280->1280 +res", M3, 1280, 1280, res=True)
    lin("L3 GEGLU 1280->10240", M3, 1280, 10240, geglu=True)
    lin("L3 ff.out 5120->1280 +res", M3, 5120, 1280, res=True)
if want("temp"):
    temp("temporal L0 320", 34, 64, 96, 320)
    temp("temporal L1 640", 34, 32, 48, 640)
    temp("temporal L2 1280", 34, 16, 24, 1280)
if want("conv"):
    conv("conv L0 320->320", 34, 64, 96, 320, 320)
    conv("conv L1 640->640", 34, 32, 48, 640, 640)
    conv("conv L2 1280->1280", 34, 16, 24, 1280, 1280)
