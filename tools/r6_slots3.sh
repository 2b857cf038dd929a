#!/bin/bash
# frames per batch x batches in flight on the target (a step stays 16 poses): images/s, live launch, frac
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd $R; mkdir -p gpurun_out
out=gpurun_out/r6_slots3.txt; : > $out
for cfg in "16 1 1" "16 1 2" "16 1 3" "8 2 2" "8 2 3" "8 2 4" "4 4 4" "4 4 6" "16 1 2" "8 2 3" "8 2 4"; do
  set -- $cfg
  RR_LANES=$(( $3 > 4 ? $3 : 4 )) python bench.py --frames-per-rank $1 --batches-per-step $2 --slots $3 --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>gpurun_out/r6_slots3.err | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('frames/batch $1 x $2, slots $3: value %.1f  live launch %.0f us  frac %.3f  chip %s' % (d['value'], r['avg_launch_us'], r['frac'], r.get('chip', {}).get('frac')))" >> $out 2>&1
done
cat $out; tail -3 gpurun_out/r6_slots3.err
