import sys, os
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
import torch
from ccedit_amd import ops
from ccedit_amd.packing import pack_weight, fold_layernorm
m, k = 208896, 320
g, b = torch.ones(320), torch.zeros(320)
gc, bc = g.cuda(), b.cuda()
for n in (320,):          # (the folded form serves single-slice layers only: ops.ln320_applicable)
    w = torch.randn(n, k) * k ** -0.5
    pw, pwl = pack_weight(w, None).to("cuda"), fold_layernorm([w], None, g, b, device="cuda")
    a = [torch.randn(m, k, device="cuda").to(torch.bfloat16) for _ in range(4)]
    def t(f):
        for x in a: f(x)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for rep in range(5):
            for x in a: f(x)
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / 20 * 1e3
    print(f"n={n}: lin320 {t(lambda x: ops.linear(x, pw)):.1f} us, layernorm {t(lambda x: ops.layernorm(x, gc, bc, 1e-5)):.1f} us, "
          f"layernorm + lin320 {t(lambda x: ops.linear(ops.layernorm(x, gc, bc, 1e-5), pw)):.1f} us, folded {t(lambda x: ops.linear(x, pwl, ln_eps=1e-5)):.1f} us")

# VERDICT r5 item 9 (GroupNorm apply, no SiLU, inside lin320s' LDS row pass in front of proj_in): what an in-LDS row pass costs is the
# difference "folded - lin320" above (the LayerNorm variant: statistics + normalise + write back + one more barrier per 32-pixel tile);
# what it would replace is the spatial GroupNorm's APPLY pass (statistics come from the producer's epilogue):
gn_g, gn_b = torch.ones(320, device="cuda"), torch.zeros(320, device="cuda")
def gn_apply(x):
    return ops.groupnorm_spatial(x.view(34, 64, 96, 320), gn_g, gn_b, 1e-6, False)
print(f"spatial GroupNorm (no SiLU) 34 x 64 x 96 x 320, statistics pass + apply pass: {t(gn_apply):.1f} us "
      f"(the apply pass alone is the `gn_spatial_apply (statistics from the producer)` row of bench.py --breakdown)")
