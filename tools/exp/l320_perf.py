"""lin320_kernel on the 64x96-level shapes, cold operands (rotating buffers behind a producer pass)."""
import os, sys, torch
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
from ccedit_amd import ops, hip
from ccedit_amd.packing import pack_weight
dev, BF = "cuda", torch.bfloat16
NB = 6
m = 208896
def run(n, res, tile=0):
    pw = pack_weight(torch.randn(n, 320) * 320 ** -0.5, torch.randn(n)).to(dev)
    src = [torch.randn(m, 320, device=dev).to(BF) for _ in range(NB)]
    xs = [torch.empty_like(s) for s in src]
    rs = [torch.randn(m, n, device=dev).to(BF) for _ in range(NB)] if res else None
    evs = []
    for rep in range(4):
        for i in range(NB):
            xs[i].copy_(src[i])
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); y = ops.linear(xs[i], pw, res1=rs[i] if res else None, tile=tile); e1.record()
            if rep: evs.append((e0, e1))
    torch.cuda.synchronize()
    ms = sum(a.elapsed_time(b) for a, b in evs) / len(evs)
    by = m * 320 * 2 + m * n * 2 * (2 if res else 1)
    print(f"{m}x{n}<-320 res={int(res)} tile={tile}: {ms * 1e3:6.1f} us  {2.0 * m * n * 320 / ms / 1e9:5.0f} TF/s  {by / ms / 1e9:5.2f} TB/s  [{hip.lib().ccedit_last_kernel().decode()}]", flush=True)
run(320, False); run(320, True); run(960, False); run(640, False)
