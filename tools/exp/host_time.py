"""Host-side enqueue time of one network evaluation (no GPU sync inside), split into ControlNet / UNet."""
import sys, os, time
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
import torch
import bench
torch.set_grad_enabled(False)
dev = torch.device("cuda")
w = bench.build_model(dev)
x, cc, cuc, hint = bench.synth_inputs(dev)
x2 = torch.cat([x, x]).contiguous()
cond = dict(crossattn=torch.cat([cuc, cc]).contiguous(), control_hint=torch.cat([hint, hint]).contiguous())
t = torch.tensor([601, 601], dtype=torch.int64, device=dev)
for _ in range(2): w(x2, t, cond)
torch.cuda.synchronize()
import ccedit_amd.network as N
cn = w.diffusion_model.controlnet
orig = cn.run
marks = {}
def timed(*a, **k):
    t0 = time.perf_counter(); r = orig(*a, **k); marks["controlnet_host_ms"] = (time.perf_counter() - t0) * 1e3; return r
cn.run = timed
for i in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter(); w(x2, t, cond); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"host enqueue {1e3*(t1-t0):.1f} ms (ControlNet part {marks['controlnet_host_ms']:.1f} ms), GPU done after {1e3*(t2-t0):.1f} ms")
