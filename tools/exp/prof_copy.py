import sys, torch
sys.path.insert(0, ".")
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
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    w(x2, t, cond)
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="count", row_limit=25, max_name_column_width=60))
evs = [e for e in prof.events() if e.name in ("aten::copy_", "aten::zero_", "aten::fill_", "aten::zeros", "aten::contiguous", "aten::clone", "aten::to")]
from collections import Counter
c = Counter()
for e in evs:
    st = [s for s in (e.stack or []) if "ccedit_amd" in s or "bench.py" in s]
    c[(e.name, tuple(st[:2]), str(e.input_shapes)[:60])] += 1
for k, v in c.most_common(25):
    print(v, k)
