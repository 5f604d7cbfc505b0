"""3x3 conv (stride 1) timing at the UNet's levels, rotating inputs; CCEDIT_HALO256 selects the rectangle size."""
import sys, os
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
import torch
from ccedit_amd import ops
from ccedit_amd.packing import pack_weight
for n, h, w, cin, cout in ((34, 64, 96, 320, 320), (34, 64, 96, 640, 320), (34, 32, 48, 640, 640), (34, 32, 48, 1280, 640), (34, 16, 24, 1280, 1280)):
    pw = pack_weight(torch.randn(cout, cin, 3, 3) * (9 * cin) ** -0.5, torch.randn(cout)).to("cuda")
    a = [torch.randn(n, h, w, cin, device="cuda").to(torch.bfloat16) for _ in range(4)]
    ref = None
    for rep in range(2):
        for x in a:
            y = ops.conv2d(x, pw)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for rep in range(5):
        for x in a:
            y = ops.conv2d(x, pw)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 20
    print(f"HALO256={os.environ.get('CCEDIT_HALO256', '1')} {n}x{h}x{w} {cin}->{cout}: {us:8.1f} us {2.0 * n * h * w * cin * 9 * cout / us / 1e6:7.1f} TF/s  checksum {y.float().abs().mean().item():.6f}")
