"""64x96-level temporal conv (320 -> 320, k3 over T) against a plain Linear of the same M, N, K: what the tap gather and the residual epilogue cost."""
import sys, os
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
import torch
from ccedit_amd import ops
from ccedit_amd.packing import pack_weight
b, t, h, w, c = 2, 17, 64, 96, 320
BF = torch.bfloat16
pwt = pack_weight(torch.randn(c, c, 3) * (3 * c) ** -0.5, torch.randn(c)).to("cuda")
pwl = pack_weight(torch.randn(c, 3 * c) * (3 * c) ** -0.5, torch.randn(c)).to("cuda")
a = [torch.randn(b * t, h, w, c, device="cuda").to(BF) for _ in range(4)]
al = [torch.randn(b * t * h * w, 3 * c, device="cuda").to(BF) for _ in range(3)]
r = [torch.randn(b * t * h * w, c, device="cuda").to(BF) for _ in range(4)]


def timeit(f, n=4):
    for i in range(n): f(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for rep in range(5):
        for i in range(n): f(i)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (5 * n)


fl = 2.0 * b * t * h * w * c * 3 * c
for tile in (1, 2, 6):
    row = []
    for name, f in (("temporal", lambda i: ops.conv_temporal(a[i], t, pwt, tile=tile)),
                    ("temporal+res", lambda i: ops.conv_temporal(a[i], t, pwt, res1=r[i], tile=tile)),
                    ("temporal+2res", lambda i: ops.conv_temporal(a[i], t, pwt, res1=r[i], res2=r[(i + 1) % 4], tile=tile)),
                    ("linear K=960", lambda i: ops.linear(al[i % 3], pwl, tile=tile)),
                    ("linear K=960+res", lambda i: ops.linear(al[i % 3], pwl, res1=r[i], tile=tile))):
        try:
            us = timeit(f)
            row.append(f"{name}: {us:6.1f} us ({fl / us / 1e6:4.0f} TF/s)")
        except Exception as e:
            row.append(f"{name}: err {str(e)[:30]}")
    print(f"tile {tile}: " + "  ".join(row), flush=True)
