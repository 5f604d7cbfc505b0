"""Per-iteration cycle counts of the fused feed-forward (CCEDIT_FF320_ABL=32 build of the kernel): mean over workgroups and waves."""
import os, sys
os.environ["CCEDIT_FF320_ABL"] = "32"
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
import torch
from ccedit_amd import ops
from ccedit_amd.packing import pack_ff320
m = 34 * 6144
g = torch.Generator().manual_seed(0)
pk = pack_ff320(torch.randn(2560, 320, generator=g) * 320 ** -0.5, torch.randn(2560, generator=g) * 0.1,
                torch.randn(320, 1280, generator=g) * 1280 ** -0.5, torch.randn(320, generator=g) * 0.1,
                1 + 0.1 * torch.randn(320, generator=g), 0.1 * torch.randn(320, generator=g), device="cuda")
x = torch.randn(m, 320, device="cuda").to(torch.bfloat16)
dbg = torch.zeros(256 * 4 * 4, dtype=torch.int64, device="cuda")
for _ in range(3):
    ops.ff320(x, pk, dbg=dbg)
torch.cuda.synchronize()
t = dbg.view(256, 4, 4).double().mean(dim=(0, 1)) / 42.0
print("s_memtime ticks per iteration: barrier A %.0f, the 60 steps %.0f, total %.0f" % (t[0], t[1], t[0] + t[1]))
