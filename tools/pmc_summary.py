#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv files: per kernel name, mean of each counter per dispatch."""
import csv, glob, os, sys, collections

def main(d):
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in files:
        with open(f) as fh:
            for r in csv.DictReader(fh):
                k = r["Kernel_Name"][:70]
                acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in acc.items():
        n = max(len(v) for v in cs.values())
        print(f"{k}  (dispatches {n})")
        for c, v in sorted(cs.items()):
            print(f"    {c:32s} mean {sum(v)/len(v):16.1f}")

if __name__ == "__main__":
    main(sys.argv[1])
