#!/bin/bash
# GPU box helper: later-pass trace grids with the segment as the fast dimension (RR_TRACE_SEG_FAST) x batches in flight
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd $R; mkdir -p gpurun_out
w=${1:-target_10M_400x200_4pass}
for i in 1 2; do for sl in 2 3 4 6; do for sf in 0 1; do
  RR_LANES=$sl RR_TRACE_SEG_FAST=$sf timeout 300 python bench.py --no-cpu-baseline --no-extras --steps 60 --warmup 5 --slots $sl --workload $w > gpurun_out/absf.log 2>&1
  echo "slots=$sl seg_fast=$sf $(grep -o "\"value\": [0-9.]*\|\"avg_launch_us\": [0-9.]*" gpurun_out/absf.log | head -2 | tr '\n' ' ')"
done; done; done
