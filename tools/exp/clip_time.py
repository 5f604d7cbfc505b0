"""Where does a clip's evaluation spend more than the step loop's?  Tight loops of wrapper calls under the clip's conditions."""
import sys, os, time
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


def loop(name, n, make_x, mark=False):
    for _ in range(3):
        w(make_x(), t, cond)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        xi = make_x()
        if mark:
            ops.set_mark(xi, '_cfg_twin_halves', True)
        w(xi, t, cond)
    torch.cuda.synchronize()
    print(f"{name:70s} {(time.perf_counter() - t0) / n * 1e3:8.2f} ms / evaluation", flush=True)


for cache in (False, True):
    w.cache_hint_stem = cache
    w.reset_caches()
    loop(f"cache={cache}: same x tensor every call", 20, lambda: x2)
    loop(f"cache={cache}: a NEW x tensor every call (device compare + host sync)", 20, lambda: x2.clone())
    ops.set_mark(t, '_cfg_twin_halves', True)
    ops.set_mark(cond["control_hint"], '_halves_equal', True)
    loop(f"cache={cache}: a NEW x tensor every call, marked (no compare)", 20, lambda: x2.clone(), mark=True)
    del t._cfg_twin_halves, cond["control_hint"]._halves_equal
    w.share_cfg_prefix = False
    loop(f"cache={cache}: share_cfg_prefix off, new x every call", 20, lambda: x2.clone())
    w.share_cfg_prefix = True
w.reset_caches()
for share in (True, False):
    w.share_cfg_prefix = share
    c = bench.time_clip(w, dev)
    print(f"clip, share_cfg_prefix={share}: sampler {c['sampler_s']} s = {c['sampler_s'] / 59 * 1e3:.2f} ms / evaluation, {c['frames_per_s']} frames/s", flush=True)
