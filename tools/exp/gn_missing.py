"""Which spatial GroupNorm sites still run their own statistics pass (no producer left statistics on the tensor)?"""
import sys, torch, collections
sys.path.insert(0, ".")
import bench
from ccedit_amd import ops
dev = torch.device("cuda")
w = bench.build_model(dev)
x, cc, cu, hint = bench.synth_inputs(dev)
x2 = torch.cat([x, x]).contiguous()
cond = dict(crossattn=torch.cat([cu, cc]).contiguous(), control_hint=torch.cat([hint, hint]).contiguous())
t = torch.tensor([601, 601], dtype=torch.int64, device=dev)
w(x2, t, cond)
orig = ops.groupnorm_spatial
miss, hit = collections.Counter(), collections.Counter()
import traceback
def spy(x, *a, **k):
    n, h, wd, c = x.shape
    if ops.gn_stats_of(x, h * wd) is None:
        fr = [f"{s.name}:{s.lineno}" for s in traceback.extract_stack()[-4:-1] if "network.py" in s.filename]
        miss[(tuple(x.shape), tuple(fr))] += 1
    else:
        hit[tuple(x.shape)] += 1
    return orig(x, *a, **k)
ops.groupnorm_spatial = spy
import ccedit_amd.network as N
w(x2, t, cond)
torch.cuda.synchronize()
print("with producer statistics:", sum(hit.values()), " own pass:", sum(miss.values()))
for k, v in sorted(miss.items(), key=lambda kv: -kv[1]):
    print(v, k)
