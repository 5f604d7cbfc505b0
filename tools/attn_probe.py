#!/usr/bin/env python3
"""Spatial attention (34 x 8 heads x 6144^2, d = 40) A/B: one subprocess per kernel variant (the switches are read once per process).
python tools/attn_probe.py            -> table;   python tools/attn_probe.py one   -> one timing in this process"""
import os, subprocess, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))

VARIANTS = {
    "general (attn_kernel)": {"CCEDIT_ATTN_SPATIAL": "0"},
    "spatial, PV in 32x32x16 tiles": {"CCEDIT_ATTN_PV16": "0"},
    "spatial, PV 32x32x16, q in log2 units": {"CCEDIT_ATTN_PV16": "0", "PROBE_Q_LOG2": "1"},
    "spatial": {},
    "spatial, q in log2 units": {"PROBE_Q_LOG2": "1"},
}


def one():
    import torch
    from ccedit_amd import ops, hip
    torch.manual_seed(0)
    n, l, heads, d = 34, 6144, 8, 40
    c = heads * d
    qkv = torch.randn(n * l, 3 * c, device="cuda").to(torch.bfloat16)
    q, k, v = qkv[:, :c], qkv[:, c:2 * c], qkv[:, 2 * c:]
    pre = os.environ.get("PROBE_Q_LOG2", "0") == "1"
    if pre:
        q = (q.float() * (d ** -0.5 * 1.4426950408889634)).to(torch.bfloat16)
    f = lambda: ops.attention(q, k, v, heads, d, batches=n, lq=l, lk=l, q_log2=pre)
    o = f()
    kern = hip.lib().ccedit_last_kernel().decode()
    # reference on one (frame, head) pair in fp32
    qq, kk, vv = (t[:l, :d].float() for t in (q, k, v))
    ref = torch.softmax(qq @ kk.T * (0.6931471805599453 if pre else d ** -0.5), -1) @ vv
    err = (o[:l, :d].float() - ref).abs().max().item()
    ts = []
    for _ in range(3):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            f()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 10)
    t = min(ts)
    print(f"{kern:28s} {t:7.3f} ms  {4.0 * n * heads * l * l * d / t / 1e9:7.1f} TF/s  max err vs fp32 {err:.2e}  (runs {[round(x, 3) for x in ts]})", flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "one":
        one()
    else:
        for rnd in range(2):
            for name, env in VARIANTS.items():
                e = dict(os.environ)
                e.update(env)
                print(f"[{name}] ", end="", flush=True)
                subprocess.run([sys.executable, __file__, "one"], env=e, check=False)
