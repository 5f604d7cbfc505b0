#!/usr/bin/env python3
"""Group a rocprofv3 --kernel-trace csv by (kernel, grid size): calls, average and total duration."""
import csv, glob, os, sys, collections
d = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else ""
acc = collections.defaultdict(lambda: [0, 0.0])
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        if pat and pat not in name:
            continue
        short = name.split("(")[0][-60:]
        key = (short, int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])), int(r.get("Grid_Size_Y", 1)))
        a = acc[key]
        a[0] += 1
        a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
for k, (n, t) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    print(f"{k[0]:62s} grid {k[1]:7d}x{k[2]:<4d} calls {n:5d} avg {t/n:9.1f} us total {t/1e3:9.3f} ms")
