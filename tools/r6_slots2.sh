#!/bin/bash
# batches in flight on the other workloads: images/s at 2 / 3 / 4 slots
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd $R; mkdir -p gpurun_out
out=gpurun_out/r6_slots2.txt; : > $out
run() {  # workload, extra args
  for s in 2 3 4; do
    python bench.py --workload $1 --slots $s $2 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('$1 slots $s: value %.1f  dominant %s live launch %.0f us  frac %.3f' % (d['value'], r['kernel'], r['avg_launch_us'], r['frac']))" >> $out
  done
}
run config2_100k_400x200_1pass ""
run config2_100k_400x200_1pass ""
run config3_1M_400x200_4pass ""
run config4_10M_400x1000_4pass ""
run config5_10M_400x1000_8pass_pertri "--frames-per-rank 1 --steps 40 --warmup 4"
run target_10M_400x200_4pass "--frames-per-rank 1"
run target_10M_400x200_4pass ""
cat $out
