# kernels whose call count scales with the number of timed steps: everything a steady-state step launches (ATen leftovers included)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for n in 4 24; do
  rm -rf /tmp/ps_$n
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ps_$n -- python $R/bench.py --steps $n --warmup 2 --no-cpu-baseline --no-clip --no-tvi2v --no-c4 > /dev/null 2>&1
  python $R/tools/prof_summary.py /tmp/ps_$n $R/gpurun_out/ps_$n.txt > /dev/null 2>&1
done
python - <<'PY'
import re,os
R=os.environ["GRAFT_REPO_ROOT"]
def load(n):
    d={}
    for l in open(f"{R}/gpurun_out/ps_{n}.txt"):
        m=re.match(r"(.{90,}?)\s+(\d+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s*$", l.rstrip("\n"))
        if m: d[m.group(1).strip()]=(int(m.group(2)), float(m.group(3)))
    return d
a,b=load(4),load(24)
rows=[]
for k,(c,t) in b.items():
    c0,t0=a.get(k,(0,0.0))
    if c>c0: rows.append(((t-t0)/20, (c-c0)/20, k))
rows.sort(reverse=True)
tot=sum(r[0] for r in rows)
print(f"per-step kernel time {tot:.2f} ms over {sum(r[1] for r in rows):.0f} launches")
for ms,cnt,k in rows: print(f"{ms:8.3f} ms {cnt:7.1f}  {k[:110]}")
PY
