#!/bin/bash
# batches in flight: images/s and the dominant kernel's live launch duration / issue fraction at 1..6 slots (target, driver's shape)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd $R; mkdir -p gpurun_out
out=gpurun_out/r6_slots.txt; : > $out
for rep in 1 2; do
for s in 1 2 3 4 5 6; do
  RR_LANES=$((s > 4 ? s : 4)) python bench.py --slots $s --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('slots $s: value %.1f  live launch %.0f us  frac %.3f  chip %s' % (d['value'], r['avg_launch_us'], r['frac'], r.get('chip', {}).get('frac')))" >> $out
done; done
cat $out
