#!/bin/bash
# GPU box helper: alternating A/B of two builds of the library (ab_libs/libradarays_{old,new}.so)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd $R; mkdir -p gpurun_out
rounds=${1:-3}; shift
W=${@:-target_10M_400x200_4pass}
for w in $W; do for i in $(seq $rounds); do for v in old new; do
  export RADARAYS_MI355_LIB=$R/ab_libs/libradarays_$v.so
  timeout 300 python bench.py --no-cpu-baseline --no-extras --steps 60 --warmup 5 --workload $w > gpurun_out/ablib_${v}_$w.log 2>&1
  echo "lib=$v $w $(grep -o "\"value\": [0-9.]*\|\"avg_launch_us\": [0-9.]*\|\"hbm_resident\": [0-9.]*" gpurun_out/ablib_${v}_$w.log | tr '\n' ' ')"
done; done; done
