run() { echo "== $1"; env $1 python bench.py --no-cpu-baseline --steps 2 --warmup 2 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d.get('clip'))"; }
run "X=1"
run "CCEDIT_ATTN_TEXT=0"
run "CCEDIT_LNF=0"
run "CCEDIT_G8_SPLIT=0"
run "CCEDIT_GRAPH=0"
