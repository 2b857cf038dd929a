#!/bin/bash
# GPU box helper: A/B of the traversal kernels (RR_TRACE_MODE 0 = quad, 1 = lane in pass 0, 2 = lane everywhere)
# usage: tools/ab_trace.sh "<modes>" [test]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd $R; mkdir -p gpurun_out
for m in ${1:-0 1 2}; do
  export RR_TRACE_MODE=$m
  if [ "$2" == "test" ]; then timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/t_mode$m.log 2>&1; echo "mode $m tests: $(grep -E 'passed|failed|rror' gpurun_out/t_mode$m.log | tail -2)"; fi
  for w in config2_100k_400x200_1pass config3_1M_400x200_4pass target_10M_400x200_4pass; do
    timeout 300 python bench.py --no-cpu-baseline --workload $w > gpurun_out/ab_${m}_$w.log 2>&1
    echo "mode $m $w $(grep -o "\"value\": [0-9.]*\|\"avg_launch_us\": [0-9.]*" gpurun_out/ab_${m}_$w.log | tr '\n' ' ')"
  done
done
