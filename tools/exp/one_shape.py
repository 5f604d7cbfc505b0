import sys, os
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
import torch
from ccedit_amd import ops, hip
from ccedit_amd.packing import pack_weight
tile = int(os.environ.get("TILE", "2")); res = int(os.environ.get("RES", "0"))
m, k, n = 208896, 320, 320
pw = pack_weight(torch.randn(n, k) * k ** -0.5, torch.randn(n)).to("cuda")
NB = 6
src = [torch.randn(m, k, device="cuda").to(torch.bfloat16) for _ in range(NB)]
a = [torch.empty_like(s) for s in src]
r = [torch.randn(m, n, device="cuda").to(torch.bfloat16) for _ in range(NB)]
g, b = torch.ones(k, device="cuda"), torch.zeros(k, device="cuda")
for rep in range(3):
    for i in range(NB):
        ops.layernorm(src[i], g, b)  # producer-like traffic
        hip.check(hip.lib().ccedit_layernorm(src[i].data_ptr(), a[i].data_ptr(), g.data_ptr(), b.data_ptr(), m, k, 1e-5, torch.cuda.current_stream().cuda_stream), "ln")
        ops.linear(a[i], pw, res1=r[i] if res else None, tile=tile)
torch.cuda.synchronize()
