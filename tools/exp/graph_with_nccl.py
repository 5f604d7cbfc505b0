"""HIP-graph capture of the evaluation with an initialised RCCL process group (its watchdog thread polls events while we capture)."""
import os, sys, torch
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
import torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
t = torch.ones(4, device="cuda"); dist.all_reduce(t); torch.cuda.synchronize()
import bench
w = bench.build_model(torch.device("cuda"))
x, cc, cu, hint = bench.synth_inputs(torch.device("cuda"))
x2 = torch.cat([x, x]).contiguous()
cond = dict(crossattn=torch.cat([cu, cc]).contiguous(), control_hint=torch.cat([hint, hint]).contiguous())
ts = torch.tensor([601, 601], dtype=torch.int64, device="cuda")
outs = []
for i in range(4):
    outs.append(w(x2, ts, cond).clone())
    dist.all_reduce(t)
torch.cuda.synchronize()
print("graph failed:", type(w)._graph_failed, "graphs:", sum("graph" in e for e in (w._graphs or {}).values()),
      "replays equal eager:", all(torch.equal(outs[0], o) for o in outs[1:]), flush=True)
dist.destroy_process_group()
