#!/bin/bash
# GPU box helper: alternating A/B of two builds of the library (ab_libs/libradarays_{old,new}.so, bench.py picks the one
# RADARAYS_MI355_LIB names).  Before the gpurun call, in the container:
#   git archive <old rev> radarays_ros_amd/csrc include | tar -x -C /tmp/old && make -C /tmp/old/radarays_ros_amd/csrc
#   mkdir -p ab_libs && cp /tmp/old/radarays_ros_amd/libradarays_mi355.so ab_libs/libradarays_old.so
#   cp radarays_ros_amd/libradarays_mi355.so ab_libs/libradarays_new.so        (*.so is git-ignored; remove ab_libs afterwards)
# usage: tools/ab_lib.sh rounds [workloads...]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd $R; mkdir -p gpurun_out
rounds=${1:-3}; shift
W=${@:-target_10M_400x200_4pass}
for w in $W; do for i in $(seq $rounds); do for v in old new; do
  export RADARAYS_MI355_LIB=$R/ab_libs/libradarays_$v.so
  timeout 300 python bench.py --no-cpu-baseline --no-extras --steps ${STEPS:-60} --warmup 5 --workload $w > gpurun_out/ablib_${v}_$w.log 2>&1
  echo "lib=$v $w $(grep -o "\"value\": [0-9.]*\|\"avg_launch_us\": [0-9.]*\|\"hbm_resident\": [0-9.]*" gpurun_out/ablib_${v}_$w.log | tr '\n' ' ')"
done; done; done
