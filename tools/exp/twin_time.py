import sys, os
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
import torch
from ccedit_amd import network
xs = [torch.randn(104448, 320, device="cuda").to(torch.bfloat16) for _ in range(4)]
for name, fn in (("twin (HIP copy)", network.twin), ("torch.cat", lambda v: torch.cat([v, v]))):
    for i in range(3):
        fn(xs[i % 4])
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(20):
        y = fn(xs[i % 4])
    e1.record()
    torch.cuda.synchronize()
    print(f"{name}: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us", torch.equal(y, torch.cat([xs[3], xs[3]])))
