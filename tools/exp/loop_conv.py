import sys, os, time
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
import torch
from ccedit_amd import ops
from ccedit_amd.packing import pack_weight
kind = os.environ.get("KIND", "conv")
if kind == "conv":
    pw = pack_weight(torch.randn(320, 320, 3, 3) * (9 * 320) ** -0.5, torch.randn(320)).to("cuda")
    a = [torch.randn(34, 64, 96, 320, device="cuda").to(torch.bfloat16) for _ in range(4)]
    f = lambda x: ops.conv2d(x, pw)
elif kind == "matmul":
    a = [torch.randn(8192, 8192, device="cuda", dtype=torch.bfloat16) for _ in range(4)]
    b = torch.randn(8192, 8192, device="cuda", dtype=torch.bfloat16)
    f = lambda x: x @ b
elif kind == "attn":
    c = 320
    a = [torch.randn(34 * 6144, 3 * c, device="cuda").to(torch.bfloat16) for _ in range(3)]
    f = lambda x: ops.attention(x[:, :c], x[:, c:2 * c], x[:, 2 * c:], 8, 40, batches=34, lq=6144, lk=6144)
elif kind == "zeros":       # same conv, all-zero data: switching activity ~0
    pw = pack_weight(torch.zeros(320, 320, 3, 3), torch.zeros(320)).to("cuda")
    a = [torch.zeros(34, 64, 96, 320, device="cuda", dtype=torch.bfloat16) for _ in range(4)]
    f = lambda x: ops.conv2d(x, pw)
t0 = time.time()
n = 0
while time.time() - t0 < 9:
    for x in a:
        f(x)
    n += len(a)
    torch.cuda.synchronize()
dt = time.time() - t0
print(f"{kind}: {dt / n * 1e6:.1f} us per call over {n} calls")
