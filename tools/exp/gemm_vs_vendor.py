"""Same setting as blaslt_ref.py (one A, one W, repeated launches: operands hot in the Infinity Cache) for ccedit_gemm with every
block shape — separates 'kernel structure' from 'cold operands in the network' when comparing with the vendor GEMM."""
import os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
import torch
from ccedit_amd import ops
from ccedit_amd.packing import pack_weight

shapes = [(8192, 8192, 8192), (208896, 320, 2880), (208896, 320, 1280), (52224, 5120, 640), (52224, 640, 2560), (52224, 640, 5760),
          (13056, 10240, 1280), (13056, 1280, 11520), (13056, 1280, 1280), (13056, 1280, 5120)]
for m, n, k in shapes:
    a = torch.randn(m, k, device="cuda", dtype=torch.bfloat16)
    pw = pack_weight(torch.randn(n, k) * k ** -0.5, torch.randn(n)).to("cuda")
    row = []
    ref = ops.linear(a, pw, tile=1).float()
    for tile in (1, 3, 4, 6, 7, 10):
        if tile in (6, 10) and n % 320:
            row.append("   -  ")
            continue
        try:
            for _ in range(3):
                c = ops.linear(a, pw, tile=tile)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                c = ops.linear(a, pw, tile=tile)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 10
            err = ((c.float() - ref).abs().max() / ref.abs().max()).item()
            row.append(f"{2.0 * m * n * k / ms / 1e9:6.0f}" + ("" if err < 2e-2 else f"(ERR {err:.1e})"))
        except Exception as e:
            row.append("  err " + str(e)[:60])
    print(f"M={m:7d} N={n:6d} K={k:6d}: TF/s by tile t1,t3,t4,t6,t7,t10: " + " ".join(row), flush=True)
