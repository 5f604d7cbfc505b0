"""Which block's output first differs between two evaluations with the ControlNet on its side stream?"""
import os, sys, torch
os.environ["CCEDIT_DEBUG_TRACE_OVERLAP"] = "1"
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
import bench
from ccedit_amd import network
dev = torch.device("cuda")
w = bench.build_model(dev)
x, cc, cu, hint = bench.synth_inputs(dev)
x2 = torch.cat([x, x]).contiguous()
cond = dict(crossattn=torch.cat([cu, cc]).contiguous(), control_hint=torch.cat([hint, hint]).contiguous())
t = torch.tensor([601, 601], dtype=torch.int64, device=dev)
w(x2, t, cond)
def traced():
    network.TRACE = {}
    y = w(x2, t, cond)
    torch.cuda.synchronize()
    tr, network.TRACE = network.TRACE, None
    return y, tr
y0, t0 = traced()
for rep in range(6):
    y1, t1 = traced()
    bad = [(k, int((t0[k].float() != t1[k].float()).sum()), t0[k].numel()) for k in t0 if not torch.equal(t0[k], t1[k])]
    print(rep, "eps equal" if torch.equal(y0, y1) else "eps DIFFERS", "| first differing traces:", bad[:6], flush=True)
