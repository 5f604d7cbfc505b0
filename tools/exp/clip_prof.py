"""cProfile of one whole clip (host side): what blocks the host beside the 59 asynchronous evaluations."""
import sys, os, cProfile, pstats, io
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
import torch
import bench
torch.set_grad_enabled(False)
dev = torch.device("cuda")
w = bench.build_model(dev)
bench.time_clip(w, dev)                      # warm: lazy packs, code objects
pr = cProfile.Profile()
pr.enable()
c = bench.time_clip(w, dev, seed=99)
pr.disable()
print(c)
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(22)
print(s.getvalue()[:5000])
