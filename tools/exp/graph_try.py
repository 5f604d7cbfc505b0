"""Experiment: capture one network evaluation in a HIP graph (torch.cuda.CUDAGraph) and replay it."""
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
def step(): return w(x2, t, cond)
for _ in range(3): out = step()
torch.cuda.synchronize()
def timeit(fn, n=5):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print("eager ms/step", timeit(step))
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(2): step()
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
try:
    with torch.cuda.graph(g):
        gout = step()
    torch.cuda.synchronize()
    print("graph ms/step", timeit(g.replay))
    ref = step()
    print("max abs diff graph vs eager", (gout - ref).abs().max().item(), "finite", torch.isfinite(gout).all().item())
except Exception as e:
    print("capture failed:", type(e).__name__, str(e)[:400])
