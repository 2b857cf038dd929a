#!/bin/bash
# GPU box helper: bench.py on torch's bundled HIP runtime (7.0.2: D2H copies are blit kernels) against the image's system
# runtime (7.2: SDMA), alternating -- RR_BENCH_SYSTEM_HIP=0 / 1
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd $R; mkdir -p gpurun_out
rounds=${1:-3}; shift
W=${@:-target_10M_400x200_4pass}
for w in $W; do for i in $(seq $rounds); do for v in 0 1; do
  RR_BENCH_SYSTEM_HIP=$v timeout 300 python bench.py --no-cpu-baseline --steps ${STEPS:-20} --warmup 5 --workload $w > gpurun_out/abrt_${v}_$w.log 2> gpurun_out/abrt_${v}_$w.err
  echo "system_hip=$v $w $(grep -o "\"value\": [0-9.]*" gpurun_out/abrt_${v}_$w.log | head -2 | tr '\n' ' ') $(grep -o '"hip_runtime": "[^"]*"' gpurun_out/abrt_${v}_$w.log) $(tail -1 gpurun_out/abrt_${v}_$w.err | cut -c1-200)"
done; done; done
