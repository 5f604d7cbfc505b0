"""Where do the device-to-device copies / fills of one network evaluation come from?  torch.profiler with Python stacks: aten ops
that launch a memcpy / fill / elementwise kernel, grouped by the innermost ccedit_amd source line."""
import os, sys, collections, torch
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
import bench
from torch.profiler import profile, ProfilerActivity
dev = torch.device("cuda")
w = bench.build_model(dev)
x, cc, cu, hint = bench.synth_inputs(dev)
x2 = torch.cat([x, x]).contiguous()
cond = dict(crossattn=torch.cat([cu, cc]).contiguous(), control_hint=torch.cat([hint, hint]).contiguous())
t = torch.tensor([601, 601], dtype=torch.int64, device=dev)
for _ in range(2):
    w(x2, t, cond)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    w(x2, t, cond)
    torch.cuda.synchronize()
by = collections.Counter()
dur = collections.Counter()
for ev in prof.events():
    if not ev.name.startswith("aten::"):
        continue
    dt = getattr(ev, "self_device_time_total", 0) or 0
    if dt <= 0:
        continue
    where = "?"
    e = ev
    while e is not None and where == "?":
        for fr in (e.stack or []):
            if "ccedit_amd" in fr or "sgm/" in fr or "bench.py" in fr:
                where = fr.split("/repo/")[-1]
                break
        e = e.cpu_parent
    by[(ev.name, where)] += 1
    dur[(ev.name, where)] += dt
tot = sum(dur.values())
print(f"aten ops with device time in one evaluation: {sum(by.values())} calls, {tot / 1e3:.2f} ms device time")
for k, n in sorted(by.items(), key=lambda kv: -dur[kv[0]])[:45]:
    print(f"{n:5d} x {dur[k] / 1e3:7.3f} ms  {k[0]:28s} {k[1]}")
