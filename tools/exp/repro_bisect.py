"""Two processes of tests/_fullsize_eval.py per policy setting: which switch makes the full-size evaluation differ from run to run?"""
import os, subprocess, sys, numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", ".."))
def run(tag, env):
    out = f"/tmp/rb_{tag}.npz"
    e = dict(os.environ); e.update(env)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_fullsize_eval.py"), out], env=e, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-1500:]
    return np.load(out)
import json
SETS = json.loads(os.environ.get("REPRO_SETS", "null")) or [["default", {}]]
for name, env in SETS:
    a, b = run(name + "_a", env), run(name + "_b", env)
    msg = []
    for k in ("eps", "eps_same", "eps_other", "frames"):
        if not np.array_equal(a[k], b[k]):
            d = np.abs(a[k].astype(np.float64) - b[k].astype(np.float64))
            msg.append(f"{k}: {int((d > 0).sum())} of {d.size} differ, max {d.max():.3g}, rel rms {np.sqrt((d**2).mean() / (b[k].astype(np.float64)**2).mean()):.3g}")
    print(name, "REPRODUCIBLE" if not msg else "; ".join(msg), flush=True)
