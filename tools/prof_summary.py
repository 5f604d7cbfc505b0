#!/usr/bin/env python3
"""Condense a rocprofv3 --kernel-trace --stats output directory into a small text summary
(per-kernel calls / total / average duration) suitable for committing under profiles/."""
import csv, glob, os, sys

def main(d, out):
    files = glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True)
    if not files:
        print("no kernel_stats.csv under", d); sys.exit(1)
    rows = []
    for f in files:
        with open(f) as fh:
            for r in csv.DictReader(fh):
                rows.append(r)
    rows.sort(key=lambda r: -float(r.get("TotalDurationNs", r.get("Total", 0)) or 0))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    with open(out, "w") as o:
        o.write(f"# source: {files}\n# total kernel time {tot/1e6:.3f} ms\n")
        o.write(f"{'kernel':90s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>10s} {'pct':>6s}\n")
        for r in rows:
            name = r["Name"][:90]
            o.write(f"{name:90s} {int(r['Calls']):7d} {float(r['TotalDurationNs'])/1e6:10.3f} {float(r['AverageNs'])/1e3:10.2f} {float(r['Percentage']):6.2f}\n")
    print(open(out).read()[:3000])

if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
