#!/bin/bash
# GPU box helper: A/B of one environment switch over the three main workloads
# usage: tools/ab_env.sh VAR "v1 v2 ..." [workloads...]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd $R; mkdir -p gpurun_out
var=$1; vals=$2; shift; shift
W=${@:-config2_100k_400x200_1pass config3_1M_400x200_4pass target_10M_400x200_4pass}
for v in $vals; do
  export $var=$v
  for w in $W; do
    timeout 300 python bench.py --no-cpu-baseline --workload $w > gpurun_out/abenv_${v}_$w.log 2>&1
    echo "$var=$v $w $(grep -o "\"value\": [0-9.]*\|\"avg_launch_us\": [0-9.]*" gpurun_out/abenv_${v}_$w.log | tr '\n' ' ')"
  done
done
