#!/bin/bash
# sample clock / power while a kernel loop runs: power_probe.sh <python script> [env...]
cd "$(dirname "$0")/../.."
python "$@" > /tmp/probe_run.log 2>&1 &
pid=$!
sleep 4
for i in 1 2 3 4 5; do
  rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Average Graphics Package Power|Current Socket Graphics Package Power|sclk clock level|Power \(W\)" | tr '\n' ';'
  echo
  sleep 0.5
done
wait $pid
tail -3 /tmp/probe_run.log
