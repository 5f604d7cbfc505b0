"""Host cost of one EAGER evaluation (the first evaluation of every clip, and the capture pass): wall per call + cProfile top."""
import sys, os, time, cProfile, pstats, io
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
import torch
import bench
torch.set_grad_enabled(False)
dev = torch.device("cuda")
w = bench.build_model(dev)
x, cc, cu, hint = bench.synth_inputs(dev, seed=43)
cond = dict(crossattn=torch.cat([cu, cc]).contiguous(), control_hint=torch.cat([hint, hint]).contiguous())
t = torch.tensor([601, 601], dtype=torch.int64, device=dev)
x2 = torch.cat([x, x]).contiguous()
w.use_graph = False
for _ in range(3):
    w(x2, t, cond)
torch.cuda.synchronize()
for name in ("eager, async host time", "eager, wall incl. GPU"):
    t0 = time.perf_counter()
    for _ in range(5):
        w(x2, t, cond)
    host = (time.perf_counter() - t0) / 5 * 1e3
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / 5 * 1e3
    print(f"eager: host enqueue {host:.1f} ms per evaluation, wall {wall:.1f} ms", flush=True)
pr = cProfile.Profile()
pr.enable()
w(x2, t, cond)
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(28)
print(s.getvalue()[:6000])
