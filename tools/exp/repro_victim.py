"""Which buffer changes AFTER it was produced (run with the ControlNet on its side stream)?  Every ops.* result is kept alive with its
checksum at production; after the step all are re-checked.  A buffer that changed was written by someone else: its nearest
lower neighbours in memory are printed (the overflowing tensor is usually the one that ends where the victim starts)."""
import os, sys, torch, collections
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
import bench
from ccedit_amd import ops, hip
dev = torch.device("cuda")
w = bench.build_model(dev)
x, cc, cu, hint = bench.synth_inputs(dev)
x2 = torch.cat([x, x]).contiguous()
cond = dict(crossattn=torch.cat([cu, cc]).contiguous(), control_hint=torch.cat([hint, hint]).contiguous())
t = torch.tensor([601, 601], dtype=torch.int64, device=dev)
w(x2, t, cond)
REG = None
def csum(y):
    v = y.reshape(-1).view(torch.int16 if y.dtype == torch.bfloat16 else (torch.int32 if y.element_size() == 4 else torch.int64)).to(torch.int64)
    return (v * (torch.arange(v.numel(), device=v.device) % 8191 + 1)).sum()
def wrap(name):
    orig = getattr(ops, name)
    def f(*a, **k):
        y = orig(*a, **k)
        if REG is not None and torch.is_tensor(y) and y.is_contiguous():
            kname = hip.lib().ccedit_last_kernel().decode() if name in ("gemm", "attention", "ff320") else ""
            REG.append([torch.cuda.current_stream().cuda_stream & 0xffff, len(REG), name, kname, tuple(y.shape), y.data_ptr(),
                        y.numel() * y.element_size(), csum(y), y])
        return y
    setattr(ops, name, f)
for n in ("gemm", "attention", "ff320", "layernorm", "groupnorm_spatial", "groupnorm_temporal", "cat_add", "add", "silu"):
    wrap(n)
ops.linear = lambda x2d, pw, **kw: ops.gemm(x2d, pw, mode=0, **kw)
yref = None
for rep in range(8):
    REG = []
    y = w(x2, t, cond)
    torch.cuda.synchronize()
    reg, REG = REG, None
    victims = [r for r in reg if int(csum(r[8])) != int(r[7])]
    if yref is None:
        yref = y.clone()
    print(f"run {rep}: {len(reg)} results kept, {len(victims)} changed after production; eps {'equal' if torch.equal(y, yref) else 'DIFFERS'}", flush=True)
    for v in victims[:4]:
        print("  VICTIM", v[:7])
        below = sorted((r for r in reg if r[5] + r[6] <= v[5]), key=lambda r: v[5] - (r[5] + r[6]))[:3]
        for b in below:
            print("     below, gap", v[5] - (b[5] + b[6]), "bytes:", b[:7])
        inside = [r for r in reg if r is not v and r[5] < v[5] + v[6] and v[5] < r[5] + r[6]]
        for b in inside[:3]:
            print("     OVERLAPS:", b[:7])
