import os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
import torch
from ccedit_amd import ops, hip
from ccedit_amd.packing import pack_weight
BF = torch.bfloat16
NB = 4
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
n = 640
pw = pack_weight(torch.randn(n, 640) * 0.04, torch.randn(n)).to("cuda")
row = os.environ.get("CCEDIT_HIP_LIB", "default").split("/")[-1] + ":"
for tiles in (1, 2, 26):
    M = 16 * 128 * tiles            # 128 pixel lanes (8 XCDs x 16) x tiles per workgroup
    xs = [torch.randn(M, 640, device="cuda").to(BF) for _ in range(NB)]
    outs = [torch.empty(M, n, dtype=BF, device="cuda") for _ in range(NB)]
    rs = [torch.randn(M, n, device="cuda").to(BF) for _ in range(NB)]
    t0 = timeit(lambda i: ops.linear(xs[i % NB], pw, out=outs[i % NB], tile=15))
    t1 = timeit(lambda i: ops.linear(xs[i % NB], pw, res1=rs[i % NB], out=outs[i % NB], tile=15))
    row += f"  {tiles:3d} tiles: plain {t0:6.1f} res {t1:6.1f} |"
    del xs, outs, rs
print(row, flush=True)
