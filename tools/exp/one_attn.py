import sys, os
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
import torch
from ccedit_amd import ops
b, heads, d, L = 34, 8, int(os.environ.get("D", "40")), int(os.environ.get("L", "6144"))
qkv = [torch.randn(b * L, 3 * heads * d, device="cuda").to(torch.bfloat16) for _ in range(3)]
c = heads * d
for rep in range(2):
    for x in qkv:
        ops.attention(x[:, :c], x[:, c:2 * c], x[:, 2 * c:], heads, d, batches=b, lq=L, lk=L)
torch.cuda.synchronize()
