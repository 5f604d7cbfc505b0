#!/usr/bin/env python3
"""Sum FETCH_SIZE / WRITE_SIZE (KB) per kernel family over one bench step.  FETCH_SIZE on gfx950 counts 64 B per
128-B request for wide streaming reads (MI355X_MICROARCH.md §HBM): the 'x2' column applies that correction."""
import csv, glob, os, sys, collections
def load(d, name):
    acc = collections.defaultdict(lambda: [0, 0.0])
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != name: continue
            k = r["Kernel_Name"]
            fam = "tap_gemm" if ("tap_gemm" in k or "g8_kernel" in k or "conv_halo" in k or "small_conv" in k or "lin320" in k or "ff320" in k) else "attn" if ("attn_kernel" in k or "attn_short" in k or "attn_spatial" in k or "attn_text" in k) else "gn_spatial_stats" if "gn_spatial_stats" in k else \
                  "gn_spatial_apply" if "gn_spatial_apply" in k else "gn_temporal" if "gn_temporal" in k else "layernorm" if "layernorm" in k else \
                  "cat_add" if "cat_add" in k else "ours_other" if "anonymous namespace" in k or "_GLOBAL__N_" in k else "torch/setup"
            acc[fam][0] += 1; acc[fam][1] += float(r["Counter_Value"])
    return acc
def load_symbols(d, name):
    """the same per kernel SYMBOL (template name without its arguments): what bench.py attaches to its dominant kernel"""
    import re
    acc = collections.defaultdict(lambda: [0, 0.0])
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != name: continue
            m = re.search(r"(\w+_kernel)\b", r["Kernel_Name"])
            if not m or ("anonymous namespace" not in r["Kernel_Name"] and "_GLOBAL__N_" not in r["Kernel_Name"]): continue
            acc[m.group(1)][0] += 1; acc[m.group(1)][1] += float(r["Counter_Value"])
    return acc
f, w = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
print(f"{'family':20s} {'launches':>8s} {'FETCH GB':>10s} {'FETCHx2 GB':>11s} {'WRITE GB':>10s}")
tf = tw = 0
for k in sorted(set(f) | set(w)):
    fb, wb = f[k][1] * 1024 / 1e9, w[k][1] * 1024 / 1e9
    if k != "torch/setup": tf += fb; tw += wb
    print(f"{k:20s} {f[k][0]:8d} {fb:10.2f} {2*fb:11.2f} {wb:10.2f}")
print(f"{'TOTAL (ours)':20s} {'':8s} {tf:10.2f} {2*tf:11.2f} {tw:10.2f}")
if os.environ.get("PMC_JSON"):
    import json as _json
    sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
    from bench import kernel_source_hash
    out = {k: {"launches": f[k][0], "fetch_bytes_x2": 2 * f[k][1] * 1024, "write_bytes": w[k][1] * 1024}
           for k in sorted(set(f) | set(w))}
    fs_, ws_ = load_symbols(sys.argv[1], "FETCH_SIZE"), load_symbols(sys.argv[2], "WRITE_SIZE")
    out["by_symbol"] = {k: {"launches": fs_[k][0], "fetch_bytes_x2": 2 * fs_[k][1] * 1024, "write_bytes": ws_[k][1] * 1024}
                        for k in sorted(set(fs_) | set(ws_))}
    out["kernel_source_hash"] = kernel_source_hash()
    _json.dump(out, open(os.environ["PMC_JSON"], "w"), indent=1)

# optional per-shape table for tap_gemm: argv[3] = shapes json dumped by bench.py --dump-shapes (same launch order)
if len(sys.argv) > 3:
    import json
    shapes = json.load(open(sys.argv[3]))
    def series(d, name):
        rows = []
        for f_ in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f_)):
                if r["Counter_Name"] == name and any(t in r["Kernel_Name"] for t in ("tap_gemm", "g8_kernel", "conv_halo", "small_conv", "lin320", "ff320")):
                    rows.append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
        rows.sort()
        return [v for _, v in rows]
    fs, ws = series(sys.argv[1], "FETCH_SIZE"), series(sys.argv[2], "WRITE_SIZE")
    n = len(shapes)
    fs, ws = fs[-n:], ws[-n:]          # the last step's launches
    agg = collections.OrderedDict()
    for sh, fv, wv in zip(shapes, fs, ws):
        key = tuple(sh[:-1])
        a = agg.setdefault(key, [0, 0.0, 0.0, 0.0])
        a[0] += 1; a[1] += 2 * fv * 1024 / 1e9; a[2] += wv * 1024 / 1e9; a[3] += sh[-1]
    print(f"\n{'shape (mode, M, N, K, stride, nres, act)':62s} {'n':>3s} {'fetch GB':>9s} {'write GB':>9s} {'alg GB':>8s} {'FLOP/B':>7s}")
    for key, (cnt, fg, wg, fl) in sorted(agg.items(), key=lambda kv: -(kv[1][1] + kv[1][2])):
        mode, M, N, K = key[0], int(key[1]), int(key[2]), int(key[3])
        nres = int(key[5])
        alg = cnt * (M * K / (9 if mode.startswith("conv") and K > 64 else 3 if mode == "temp" else 1) * 2 + M * N * 2 * (1 + nres) + N * K * 2) / 1e9
        print(f"{str(key):62s} {cnt:3d} {fg:9.2f} {wg:9.2f} {alg:8.2f} {fl/((fg+wg)*1e9+1):7.0f}")
