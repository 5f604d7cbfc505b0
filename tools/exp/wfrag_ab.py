#!/usr/bin/env python3
"""CcGemmDesc.Wfrag (fragment-ordered weight copy) on / off for the three register-resident-weight kernels: identical values, time per
launch with cold operands (graph replay of NB rotating buffers)."""
import os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
import torch
from ccedit_amd import ops, hip
from ccedit_amd.packing import pack_weight, fold_layernorm
BF = torch.bfloat16
NB = 5


def timeit(f, n=20):
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


def ab(name, fn):
    out = []
    for on in (True, False, True, False):
        ops.WFRAG = on
        out.append(timeit(fn))
    ops.WFRAG = True
    y1 = fn(0)
    k = hip.lib().ccedit_last_kernel().decode()
    ops.WFRAG = False
    y0 = fn(0)
    ops.WFRAG = True
    print(f"{name:44s} wfrag {out[0]:6.1f} {out[2]:6.1f} us | row-major {out[1]:6.1f} {out[3]:6.1f} us | identical {torch.equal(y0, y1)}  [{k}]", flush=True)


torch.manual_seed(0)
M = 34 * 1536
xs = [torch.randn(M, 640, device="cuda").to(BF) for _ in range(NB)]
for n in (640, 1920):
    pw = pack_weight(torch.randn(n, 640) * 0.04, torch.randn(n)).to("cuda")
    rs = [torch.randn(M, n, device="cuda").to(BF) for _ in range(NB)]
    ab(f"lin 52224 x {n} <- 640", lambda i: ops.linear(xs[i % NB], pw, tile=10))
    ab(f"lin 52224 x {n} <- 640 + res", lambda i: ops.linear(xs[i % NB], pw, res1=rs[i % NB], tile=10))
    del rs
del xs
M = 34 * 6144
xs = [torch.randn(M, 320, device="cuda").to(BF) for _ in range(NB)]
for n in (320, 960):
    pw = pack_weight(torch.randn(n, 320) * 0.05, torch.randn(n)).to("cuda")
    ab(f"lin 208896 x {n} <- 320", lambda i: ops.linear(xs[i % NB], pw))
rs = [torch.randn(M, 320, device="cuda").to(BF) for _ in range(NB)]
pw = pack_weight(torch.randn(320, 320) * 0.05, torch.randn(320)).to("cuda")
ab("lin 208896 x 320 <- 320 + res", lambda i: ops.linear(xs[i % NB], pw, res1=rs[i % NB]))
g, be = torch.randn(320) * 0.2 + 1, torch.randn(320) * 0.2
pl = fold_layernorm([torch.randn(320, 320) * 0.05], [torch.randn(320)], g, be).to("cuda")
ab("lin 208896 x 320 <- LN(320)", lambda i: ops.linear(xs[i % NB], pl, ln_eps=1e-5))
pt = pack_weight(torch.randn(320, 320, 3) * 0.03, torch.randn(320)).to("cuda")
x4 = [x.view(34, 64, 96, 320) for x in xs]
ab("temp 34 x 64 x 96 x 320 k3 + res", lambda i: ops.conv_temporal(x4[i % NB], 17, pt, res1=rs[i % NB]))
