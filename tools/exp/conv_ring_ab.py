"""3x3 stride-1 convs of the 64x96 / 32x48 levels: policy conv_halo = 2 (two-slot weight ring, one barrier per tap) against 1 (four-slot
ring of K = 32 half tiles, counted waits), same process, alternating, rotating inputs; the two must give the same bits (same k-step order)."""
import sys, os
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
import torch
from ccedit_amd import hip, ops
from ccedit_amd.packing import pack_weight
lib = hip.lib()
tot = {1: 0.0, 2: 0.0}      # key: 1 = two-slot (policy value 2), 2 = four-slot (policy value 1)
for n, h, w, cin, cout, cnt in ((34, 64, 96, 320, 320, 14), (34, 64, 96, 640, 320, 2), (34, 64, 96, 960, 320, 1), (34, 32, 48, 320, 640, 2),
                                (34, 32, 48, 640, 640, 9), (34, 32, 48, 960, 640, 1), (34, 32, 48, 1280, 640, 1), (34, 32, 48, 1920, 640, 1),
                                (17, 64, 96, 320, 320, 0), (34, 8, 12, 1280, 320, 0)):
    pw = pack_weight(torch.randn(cout, cin, 3, 3) * (9 * cin) ** -0.5, torch.randn(cout)).to("cuda")
    a = [torch.randn(n, h, w, cin, device="cuda").to(torch.bfloat16) for _ in range(3)]
    res, outs = {}, {}
    for rnd in range(3):
        for pol in (1, 2):
            lib.ccedit_policy_set(b"conv_halo", 3 - pol)
            for x in a:
                y = ops.conv2d(x, pw)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for rep in range(4):
                for x in a:
                    y = ops.conv2d(x, pw)
            e1.record()
            torch.cuda.synchronize()
            res.setdefault(pol, []).append(e0.elapsed_time(e1) * 1e3 / 12)
            outs[pol] = y
    lib.ccedit_policy_set(b"conv_halo", 1)
    same = torch.equal(outs[1], outs[2])
    fl = 2.0 * n * h * w * cin * 9 * cout
    b1, b2 = min(res[1]), min(res[2])
    tot[1] += cnt * b1; tot[2] += cnt * b2
    print(f"{n}x{h}x{w} {cin}->{cout}: two-slot {b1:8.1f} us {fl / b1 / 1e6:7.1f} TF/s | four-slot {b2:8.1f} us {fl / b2 / 1e6:7.1f} TF/s | ratio {b2 / b1:.4f} | same bits {same} | {lib.ccedit_last_kernel().decode()}")
print(f"weighted by launches per step: two-slot {tot[1] / 1e3:.2f} ms, four-slot {tot[2] / 1e3:.2f} ms")
