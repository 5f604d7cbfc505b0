"""In-process reproducibility probe at full size: the same network evaluation N times, bitwise comparison with the first."""
import os, sys, torch
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
import bench
dev = torch.device("cuda")
w = bench.build_model(dev)
x, cc, cu, hint = bench.synth_inputs(dev)
x2 = torch.cat([x, x]).contiguous()
cond = dict(crossattn=torch.cat([cu, cc]).contiguous(), control_hint=torch.cat([hint, hint]).contiguous())
t = torch.tensor([601, 601], dtype=torch.int64, device=dev)
ref = w(x2, t, cond).clone()
bad = []
for i in range(int(os.environ.get("REPS", "6"))):
    y = w(x2, t, cond)
    if not torch.equal(y, ref):
        d = (y - ref).abs()
        bad.append((i, int((d > 0).sum()), [int((d[h] > 0).sum()) for h in range(2)], float(d.max())))
print(os.environ.get("TAG", ""), "REPRODUCIBLE" if not bad else f"DIFFERS {bad}", flush=True)
