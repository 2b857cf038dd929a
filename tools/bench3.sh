#!/bin/bash
# GPU box helper: parity tests + the three bench workloads, one line each
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd $R; mkdir -p gpurun_out
if [ "$1" != "notest" ]; then timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/t.log 2>&1; grep -E "passed|failed|rror" gpurun_out/t.log | tail -5; fi
for w in config2_100k_400x200_1pass config3_1M_400x200_4pass target_10M_400x200_4pass; do
  timeout 300 python bench.py --no-cpu-baseline --workload $w > gpurun_out/b_$w.log 2>&1
  echo $w $(grep -o "\"value\": [0-9.]*\|\"avg_launch_us\": [0-9.]*" gpurun_out/b_$w.log)
done
