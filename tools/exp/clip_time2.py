"""Host-side pacing of the evaluations of one clip: time between consecutive network calls."""
import sys, os, time
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
import torch
import bench
torch.set_grad_enabled(False)
dev = torch.device("cuda")
w = bench.build_model(dev)
stamps = []
orig = type(w).forward


def fwd(self, x, t, c, **kw):
    stamps.append(time.perf_counter())
    return orig(self, x, t, c, **kw)


type(w).forward = fwd
for sync_each in (False, True):
    stamps.clear()
    if sync_each:
        def fwd2(self, x, t, c, **kw):
            torch.cuda.synchronize()
            stamps.append(time.perf_counter())
            return orig(self, x, t, c, **kw)
        type(w).forward = fwd2
    c = bench.time_clip(w, dev)
    d = [(b - a) * 1e3 for a, b in zip(stamps, stamps[1:])]
    print(f"sync_each={sync_each}: sampler {c['sampler_s']} s; first intervals {[round(v, 1) for v in d[:5]]}; mean of the rest {sum(d[5:]) / len(d[5:]):.2f} ms; "
          f"min {min(d[5:]):.2f} max {max(d[5:]):.2f}", flush=True)
