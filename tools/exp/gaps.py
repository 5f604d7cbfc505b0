"""Idle time between consecutive kernels of the steady-state steps of a rocprofv3 --kernel-trace run of bench.py (single stream):
how much of a step is launch gaps?   python tools/exp/gaps.py <trace dir>
NOTE (round 5): the LAST evaluation of a bench.py run is the event-bracketed profiled step behind `by_kernel` — eager, a HIP event pair
around every launch, ~10.5 us of idle after each kernel (the '|' rows of the pattern printed at the end).  The graph-replayed timed
steps before it show NO gaps (consecutive kernels overlap by the dispatch pipeline: '.'), i.e. nothing is left to win from launch
overhead on the single-GPU path."""
import csv, glob, os, sys
files = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)
rows = []
for f in files:
    with open(f) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
# the timed steps: the last 5 evaluations; an evaluation starts with the ncthw_to_nhwc of x (first kernel after a long idle gap)
tail = rows[-3000:]
busy = sum(e - s for s, e, _ in tail)
span = tail[-1][1] - tail[0][0]
gaps = [(tail[i + 1][0] - tail[i][1], tail[i][2][:60], tail[i + 1][2][:60]) for i in range(len(tail) - 1)]
pos = [g for g in gaps if g[0] > 0]
print(f"{len(tail)} kernels: span {span / 1e6:.2f} ms, busy {busy / 1e6:.2f} ms, idle {sum(g[0] for g in pos) / 1e6:.2f} ms in {len(pos)} gaps "
      f"(median {sorted(g[0] for g in pos)[len(pos) // 2] / 1e3:.2f} us); overlapped pairs {len(gaps) - len(pos)}")
import collections
hist = collections.Counter(min(int(g[0] / 1000), 20) for g in pos)
print("gap histogram (us: count):", dict(sorted(hist.items())))
big = sorted(pos, reverse=True)[:12]
for g in big:
    print(f"  {g[0] / 1e3:8.1f} us  after {g[1]}  before {g[2]}")
by = collections.Counter()
for g in pos:
    by[g[1].split("(")[0][-40:]] += g[0]
print("idle by preceding kernel (ms):", [(k, round(v / 1e6, 3)) for k, v in by.most_common(10)])
# the ~10 us class: which (previous, next) kernel pairs carry it?
mid = [g for g in pos if 8000 <= g[0] <= 12000]
pairs = collections.Counter((g[1].replace("void (anonymous namespace)::", "")[:44], g[2].replace("void (anonymous namespace)::", "")[:44]) for g in mid)
print(f"{len(mid)} gaps of 8-12 us, by (previous -> next):")
for (a_, b_), n in pairs.most_common(40):
    print(f"  {n:4d}  {a_:44s} -> {b_}")
nxt = collections.Counter(g[2].replace("void (anonymous namespace)::", "")[:50] for g in mid)
print("by NEXT kernel:", nxt.most_common(25))
prv = collections.Counter(g[1].replace("void (anonymous namespace)::", "")[:50] for g in mid)
print("by PREVIOUS kernel:", prv.most_common(25))
# where in the launch sequence do they fall?  one character per kernel of the last ~700 launches: '|' = followed by an 8-12 us gap
seq = "".join("|" if 8000 <= g[0] <= 12000 else ("." if g[0] <= 0 else "o") for g in gaps[-760:])
for i in range(0, len(seq), 120):
    print(seq[i:i + 120])
names = [t[2].replace("void (anonymous namespace)::", "")[:28] for t in tail[-760:]]
durs = [(t[1] - t[0]) / 1e3 for t in tail[-760:]]
print("first 60 launches of that window: name, duration us, gap after (us)")
for i in range(60):
    print(f"  {names[i]:28s} {durs[i]:8.1f} {gaps[-760:][i][0] / 1e3:8.1f}")
