"""Linears / temporal convs with residual epilogues: us per launch."""
import sys, os
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
import torch
from ccedit_amd import ops
from ccedit_amd.packing import pack_weight
torch.manual_seed(0)
def t(f, n=20):
    for _ in range(3): f(0)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n): f(i % 4)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for m, n, k in ((52224, 640, 640), (13056, 1280, 1280), (52224, 640, 2560), (208896, 320, 960)):
    pw = pack_weight(torch.randn(n, k) * k ** -0.5, torch.randn(n)).to("cuda")
    a = [torch.randn(m, k, device="cuda").to(torch.bfloat16) for _ in range(4)]
    r = [torch.randn(m, n, device="cuda").to(torch.bfloat16) for _ in range(4)]
    y = ops.linear(a[0], pw, res1=r[0])
    print(f"lin {m}x{n}x{k}: plain {t(lambda i: ops.linear(a[i], pw)):.1f} us, +res {t(lambda i: ops.linear(a[i], pw, res1=r[i])):.1f} us, +2res {t(lambda i: ops.linear(a[i], pw, res1=r[i], res2=r[(i+1)%4])):.1f} us  chk {y.float().double().sum().item():.3f}")
b, T = 2, 17
for h, w, c in ((64, 96, 320), (32, 48, 640), (16, 24, 1280)):
    pw = pack_weight(torch.randn(c, c, 3) * (3 * c) ** -0.5, torch.randn(c)).to("cuda")
    a = [torch.randn(b * T, h, w, c, device="cuda").to(torch.bfloat16) for _ in range(4)]
    r = [torch.randn(b * T * h * w, c, device="cuda").to(torch.bfloat16) for _ in range(4)]
    e = torch.randn(b, c, device="cuda")
    y = ops.conv_temporal(a[0], T, pw, res1=r[0], group_bias=e, group_rows=T * h * w)
    print(f"temporal {h}x{w} C={c}: plain {t(lambda i: ops.conv_temporal(a[i], T, pw)):.1f} us, +res {t(lambda i: ops.conv_temporal(a[i], T, pw, res1=r[i])):.1f} us, +res+emb {t(lambda i: ops.conv_temporal(a[i], T, pw, res1=r[i], group_bias=e, group_rows=T * h * w)):.1f} us, +2res {t(lambda i: ops.conv_temporal(a[i], T, pw, res1=r[i], res2=r[(i + 1) % 4])):.1f} us  chk {y.float().double().sum().item():.3f}")
