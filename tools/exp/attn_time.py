"""Spatial self-attention at the 64x96 / 32x48 levels: us per launch, TF/s, for the kernel selected by CCEDIT_ATTN_PP."""
import sys, os
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
import torch
from ccedit_amd import ops
for d, L in ((40, 6144), (80, 1536)):
    b, heads = 34, 8
    c = heads * d
    qkv = [torch.randn(b * L, 3 * c, device="cuda").to(torch.bfloat16) for _ in range(3)]
    for x in qkv:
        ops.attention(x[:, :c], x[:, c:2 * c], x[:, 2 * c:], heads, d, batches=b, lq=L, lk=L)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for rep in range(4):
        for x in qkv:
            ops.attention(x[:, :c], x[:, c:2 * c], x[:, 2 * c:], heads, d, batches=b, lq=L, lk=L)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 12
    print(f"PP={os.environ.get('CCEDIT_ATTN_PP', '1')} d={d} L={L}: {us:9.1f} us  {4.0 * b * heads * L * L * d / us / 1e6:7.1f} TF/s")
