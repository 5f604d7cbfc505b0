import sys, os
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
import torch
from ccedit_amd import ops
from ccedit_amd.packing import pack_weight
tile = int(os.environ.get("TILE", "9"))
m, k = 208896, 320
NB = 4
for n, geglu, res in ((320, False, False), (320, False, True), (1280, True, False), (960, False, False), (640, False, False)):
    w = torch.randn(2 * n if geglu else n, k) * k ** -0.5
    pw = pack_weight(w, torch.randn(w.shape[0]), geglu=geglu).to("cuda")
    a = [torch.randn(m, k, device="cuda").to(torch.bfloat16) for _ in range(NB)]
    r = [torch.randn(m, n, device="cuda").to(torch.bfloat16) for _ in range(NB)]
    o = [torch.empty(m, n, device="cuda", dtype=torch.bfloat16) for _ in range(NB)]
    for i in range(NB):
        ops.linear(a[i], pw, res1=r[i] if res else None, out=o[i], tile=tile)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for rep in range(5):
        for i in range(NB):
            ops.linear(a[i], pw, res1=r[i] if res else None, out=o[i], tile=tile)
    e1.record()
    torch.cuda.synchronize()
    print(f"tile {tile} n={n} geglu={geglu} res={res}: {e0.elapsed_time(e1) / (5 * NB) * 1e3:.1f} us")
