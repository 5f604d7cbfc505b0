#!/bin/bash
# Everything profiles/ holds for one round, in one call on the GPU box:  tools/profile_round.sh <tag>   (e.g. r02)
#   <tag>_bench_line_default.json       python bench.py                                   (the driver's command, clip included)
#   <tag>_bench_line_tvi2v.json         python bench.py --workload tvi2v
#   <tag>_bench_line_under_rocprof.json + <tag>_bench_kernel_stats.txt        rocprofv3 --kernel-trace --stats, batched single stream
#   <tag>_bench_line_under_rocprof_streams.json + <tag>_bench_kernel_stats_streams.txt   the same, default multi-stream execution
#   <tag>_vae_fp32_time.txt + <tag>_vae_fp32_kernel_stats.txt   full-size first-stage decode, bf16 default against the fp32 option (tools/vae32_time.py)
#   <tag>_pmc_traffic.json / .txt       FETCH_SIZE / WRITE_SIZE passes (tools/pmc_traffic.sh), tagged with the kernel-source hash
# Outputs land in gpurun_out/; copy what is to be judged into profiles/.
tag=${1:-rXX}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python $R/bench.py --json-out $O/${tag}_bench_line_default.json > /dev/null 2>&1
if [ "${PROFILE_TVI2V:-0}" = 1 ]; then python $R/bench.py --workload tvi2v 2>/dev/null | tail -1 > $O/${tag}_bench_line_tvi2v.json; fi   # (the default line carries a tvi2v object)
for mode in single streams; do
  rm -rf /tmp/pf_$mode
  if [ $mode = single ]; then export CCEDIT_SPLIT_CFG=0 CCEDIT_OVERLAP_CONTROLNET=0; sfx=""; else unset CCEDIT_SPLIT_CFG CCEDIT_OVERLAP_CONTROLNET; sfx="_streams"; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pf_$mode -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-clip --no-tvi2v 2>/dev/null | tail -1 > $O/${tag}_bench_line_under_rocprof$sfx.json
  python $R/tools/prof_summary.py /tmp/pf_$mode $O/${tag}_bench_kernel_stats$sfx.txt > /dev/null 2>&1
done
unset CCEDIT_SPLIT_CFG CCEDIT_OVERLAP_CONTROLNET
rm -rf /tmp/pf_vae32
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pf_vae32 -- python $R/tools/vae32_time.py > $O/${tag}_vae_fp32_time.txt 2>/dev/null
python $R/tools/prof_summary.py /tmp/pf_vae32 $O/${tag}_vae_fp32_kernel_stats.txt > /dev/null 2>&1
PMC_JSON=$O/${tag}_pmc_traffic.json bash $R/tools/pmc_traffic.sh > $O/${tag}_pmc_traffic.txt 2>&1
if [ "${PROFILE_TVI2V:-0}" = 1 ]; then PMC_BENCH_ARGS="--workload tvi2v" PMC_JSON=$O/${tag}_pmc_traffic_tvi2v.json bash $R/tools/pmc_traffic.sh > $O/${tag}_pmc_traffic_tvi2v.txt 2>&1; fi
# matrix-pipe / VALU counters of the dominant kernels (tools/pmc_counters.sh: two --pmc passes each, --kernel-trace only); round 5: the feed-forward alone and as the block tail;
# round 6: the spatial attention with the optimistic reference against the tracked one, the 3x3 conv with the four-slot weight ring against the two-slot one,
# the fp32 first stage's 3x3 conv as six bf16 products (f32p / f32s) against the fp32 matrix instruction
rm -f $O/${tag}_pmc_counters.txt
for spec in "f32p_gemm|f32conv|" "f32_gemm|f32conv|f32_split=0" "f32s_gemm|f32conv128|" "ff320|ff320|" "ff320|ff320tail|" "attn_spatial|attnq|" "attn_spatial|attnq|attn_opt=0" "attn_spatial|attnq80|" "attn_spatial|attnq80|attn_opt=0" "conv_halo|conv|" "conv_halo|conv|conv_halo=2" "g8_kernel|g8geglu|"; do
  IFS='|' read -r pat case pol <<< "$spec"
  echo "=== $case ($pat) CCEDIT_POLICY='$pol' ===" >> $O/${tag}_pmc_counters.txt
  CCEDIT_POLICY="$pol" bash $R/tools/pmc_counters.sh $pat $case >> $O/${tag}_pmc_counters.txt 2>&1
done
# config 4 as a FUNCTIONAL record on this one-GPU box: two ranks time-slicing the GPU through host-staged gloo (gpus_physical = 1 on the line)
CCEDIT_DIST_BACKEND=gloo python $R/bench.py --gpus 2 --steps 2 --warmup 1 --no-cpu-baseline --no-clip --no-tvi2v --no-profile-step --c4-deadline 900 --json-out $O/${tag}_bench_line_rows_gloo2.json > /dev/null 2>&1
# one measured full-size oracle evaluation on the host beside the crop conversion
if [ "${PROFILE_CPU_FULL:-1}" = 1 ]; then python $R/bench.py --steps 3 --warmup 2 --no-clip --no-tvi2v --cpu-full-step --json-out $O/${tag}_bench_line_cpu_full_step.json > /dev/null 2>&1; fi
ls -la $O/${tag}_*
