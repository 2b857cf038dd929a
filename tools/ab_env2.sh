#!/bin/bash
# GPU box helper: ALTERNATING A/B of one environment switch (box clocks drift: interleave the runs)
# usage: tools/ab_env2.sh VAR "v1 v2" rounds [workloads...]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd $R; mkdir -p gpurun_out
var=$1; vals=$2; rounds=$3; shift; shift; shift
W=${@:-target_10M_400x200_4pass}
for w in $W; do for i in $(seq $rounds); do for v in $vals; do
  export $var=$v
  timeout 300 python bench.py --no-cpu-baseline --no-extras --steps 60 --warmup 5 --workload $w > gpurun_out/ab2_${var}_${v}_$w.log 2>&1
  echo "$var=$v $w $(grep -o "\"value\": [0-9.]*\|\"avg_launch_us\": [0-9.]*" gpurun_out/ab2_${var}_${v}_$w.log | tr '\n' ' ')"
done; done; done
