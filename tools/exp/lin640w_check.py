#!/usr/bin/env python3
"""lin640w_kernel (tile 15) against lin640s_kernel (tile 10) / the automatic dispatch: values and time, cold operands."""
import os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
import torch
from ccedit_amd import ops, hip
from ccedit_amd.packing import pack_weight, fold_layernorm
BF = torch.bfloat16
NB = 5
TILES = [int(t) for t in os.environ.get("TILES", "15,10,0").split(",")]


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


torch.manual_seed(0)
for M in (16 * 7, 16 * 333, 34 * 1536):
    for n in (320, 640, 1920):
        x = torch.randn(M, 640, device="cuda").to(BF)
        w, b = torch.randn(n, 640) * 0.04, torch.randn(n)
        pw = pack_weight(w, b).to("cuda")
        r = torch.randn(M, n, device="cuda").to(BF)
        ref = torch.nn.functional.linear(x.float(), w.cuda(), b.cuda())
        y = ops.linear(x, pw, tile=15)
        k = hip.lib().ccedit_last_kernel().decode()
        e0 = (y.float() - ref).abs().max().item()
        yr = ops.linear(x, pw, res1=r, tile=15)
        e1 = (yr.float() - ref - r.float()).abs().max().item()
        y2 = ops.linear(x, pw, res1=r, row_sums=True, tile=15)
        sums = ops.ln_sums_of(y2)
        y2d = y2.double()
        e2 = max((sums[:, 0] - y2d.sum(1)).abs().max().item(), ((sums[:, 1] - (y2d ** 2).sum(1)).abs() / (1 + (y2d ** 2).sum(1))).max().item())
        same = torch.equal(y2, yr) and torch.equal(ops.linear(x, pw, res1=r, tile=15), yr)
        g, be = torch.randn(640) * 0.2 + 1, torch.randn(640) * 0.2
        pl = fold_layernorm([w], [b], g, be).to("cuda")
        st = ops.row_stats(x, 1e-5)
        refl = torch.nn.functional.linear(torch.nn.functional.layer_norm(x.float(), (640,), g.cuda(), be.cuda(), 1e-5), w.cuda(), b.cuda())
        yl = ops.linear(x, pl, ln_stats=st, tile=15)
        e3 = (yl.float() - refl).abs().max().item()
        sx = torch.stack([x.double().sum(1), (x.double() ** 2).sum(1)], 1)
        yl2 = ops.linear(x, pl, ln_sums=(sx, 1e-5), tile=15)
        e4 = (yl2.float() - refl).abs().max().item()
        y10 = ops.linear(x, pw, res1=r, tile=10) if n % 128 == 0 else yr
        print(f"M={M:6d} N={n:4d} [{k}] plain {e0:.3e} res {e1:.3e} sums {e2:.2e} same={same} ln {e3:.3e} ln_sums {e4:.3e}  vs tile10 {(y10.float() - yr.float()).abs().max().item():.3e}  (ref scale {ref.abs().max().item():.2f})", flush=True)
M = 34 * 1536
xs = [torch.randn(M, 640, device="cuda").to(BF) for _ in range(NB)]
sts = [ops.row_stats(x, 1e-5) for x in xs]
for n in (640, 1280, 1920):
    rs = [torch.randn(M, n, device="cuda").to(BF) for _ in range(NB)]
    pw = pack_weight(torch.randn(n, 640) * 0.04, torch.randn(n)).to("cuda")
    pl = fold_layernorm([torch.randn(n, 640) * 0.04], [torch.randn(n)], torch.ones(640), torch.zeros(640)).to("cuda")
    outs = [torch.empty(M, n, dtype=BF, device="cuda") for _ in range(NB)]
    fl = 2 * M * 640 * n
    for tile in TILES:
        t0 = timeit(lambda i: ops.linear(xs[i % NB], pw, out=outs[i % NB], tile=tile))
        k = hip.lib().ccedit_last_kernel().decode()
        t1 = timeit(lambda i: ops.linear(xs[i % NB], pw, res1=rs[i % NB], out=outs[i % NB], tile=tile))
        t2 = timeit(lambda i: ops.linear(xs[i % NB], pw, res1=rs[i % NB], out=outs[i % NB], row_sums=True, tile=tile))
        t3 = timeit(lambda i: ops.linear(xs[i % NB], pl, ln_stats=sts[i % NB], out=outs[i % NB], tile=tile))
        k3 = hip.lib().ccedit_last_kernel().decode()
        print(f"N={n:4d} tile {tile:2d}: plain {t0:6.1f} us {fl / t0 / 1e6:5.0f} TF | res {t1:6.1f} {fl / t1 / 1e6:5.0f} | res+sums {t2:6.1f} | ln {t3:6.1f} {fl / t3 / 1e6:5.0f}  [{k} / {k3}]", flush=True)
    del rs, outs
