"""Spatial self-attention at the 64x96 / 32x48 levels, same box, same process: policy attn_opt = 1 (reference fixed by the first key
tile, exact re-run on overflow) against 0 (reference tracked on every tile), alternating, rotating inputs; q in log2 units."""
import sys, os
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
import torch
from ccedit_amd import hip, ops
lib = hip.lib()
for d, L in ((40, 6144), (80, 1536)):
    b, heads = 34, 8
    c = heads * d
    qkv = [torch.randn(b * L, 3 * c, device="cuda").to(torch.bfloat16) for _ in range(3)]
    for x in qkv:
        x[:, :c] *= d ** -0.5 * 1.4426950408889634
    res = {}
    for rnd in range(3):
        for opt in (1, 0):
            lib.ccedit_policy_set(b"attn_opt", opt)
            for x in qkv:
                o = ops.attention(x[:, :c], x[:, c:2 * c], x[:, 2 * c:], heads, d, batches=b, lq=L, lk=L, q_log2=True)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for rep in range(4):
                for x in qkv:
                    o = ops.attention(x[:, :c], x[:, c:2 * c], x[:, 2 * c:], heads, d, batches=b, lq=L, lk=L, q_log2=True)
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / 12
            res.setdefault(opt, []).append(us)
            print(f"d={d} L={L} attn_opt={opt} round {rnd}: {us:9.1f} us  {4.0 * b * heads * L * L * d / us / 1e6:7.1f} TF/s  {lib.ccedit_last_kernel().decode()} finite={bool(torch.isfinite(o.float()).all())}")
    lib.ccedit_policy_set(b"attn_opt", 1)
    print(f"d={d}: best opt {min(res[1]):.1f} us, best tracked {min(res[0]):.1f} us, ratio {min(res[1]) / min(res[0]):.4f}")
