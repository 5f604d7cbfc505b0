"""Repeat the d = 80 (and d = 40) spatial attention launch and compare the outputs bit for bit: which policy arm differs between launches,
and where (whole 256-query workgroup blocks = the exact re-run was taken, single rows = a race)."""
import sys, os
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
import torch
from ccedit_amd import hip, ops
lib = hip.lib()
BF = torch.bfloat16
for d, heads, lq, lk in ((80, 4, 1536, 1536), (40, 8, 1536, 1536), (80, 8, 1536, 1536), (40, 8, 6144, 6144)):
    b = 2 if lq < 4000 else 4
    c = heads * d
    g = torch.Generator().manual_seed(1)
    q, k, v = (torch.randn(b * l, c, generator=g).to(BF).cuda() for l in (lq, lk, lk))
    for opt in (1, 0):
        lib.ccedit_policy_set(b"attn_opt", opt)
        ref = ops.attention(q, k, v, heads, d, batches=b, lq=lq, lk=lk).clone()
        ndiff, rows_bad = 0, set()
        for it in range(200):
            o = ops.attention(q, k, v, heads, d, batches=b, lq=lq, lk=lk)
            if not torch.equal(o, ref):
                ndiff += 1
                bad = (o != ref).view(b, lq, heads, d).any(-1).nonzero()
                for bb, r, h in bad.tolist()[:2000]:
                    rows_bad.add((bb, h, r // 256))
                if ndiff <= 2:
                    rr = (o != ref).view(b, lq, heads, d).any(-1)
                    print(f"   d={d} opt={opt} iter {it}: {int(rr.sum())} (row, head) cells differ; per (batch, head) counts: {rr.sum(1).tolist()}; "
                          f"max |diff| {float((o.float() - ref.float()).abs().max()):.3e}")
        print(f"d={d} heads={heads} {lq}x{lk} attn_opt={opt}: {ndiff} of 200 launches differ from the first; (batch, head, 256-row block) touched: {sorted(rows_bad)[:12]}")
lib.ccedit_policy_set(b"attn_opt", 1)
