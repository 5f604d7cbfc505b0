"""Temporal conv (k3 over T) at the UNet levels: us per launch by block shape, rotating inputs, with the residual epilogue."""
import sys, os
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
import torch
from ccedit_amd import ops
from ccedit_amd.packing import pack_weight
b, t = 2, 17
for h, w, c in ((64, 96, 320), (32, 48, 640), (16, 24, 1280)):
    pw = pack_weight(torch.randn(c, c, 3) * (3 * c) ** -0.5, torch.randn(c)).to("cuda")
    a = [torch.randn(b * t, h, w, c, device="cuda").to(torch.bfloat16) for _ in range(4)]
    r = [torch.randn(b * t * h * w, c, device="cuda").to(torch.bfloat16) for _ in range(4)]
    row = []
    for tile in (1, 1, 2, 12, 13, 6):
        try:
            for i in range(4):
                ops.conv_temporal(a[i], t, pw, res1=r[i], tile=tile)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for rep in range(5):
                for i in range(4):
                    ops.conv_temporal(a[i], t, pw, res1=r[i], tile=tile)
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / 20
            row.append(f"t{tile}: {us:7.1f} us ({2.0 * b * t * h * w * c * 3 * c / us / 1e6:5.0f} TF/s)")
        except Exception as e:
            row.append(f"t{tile}: err {str(e)[:40]}")
    print(f"{h}x{w} C={c}: " + "  ".join(row), flush=True)
