#!/bin/bash
# A/B of two builds of the library over the bench workloads: tools/ab_lib.sh <other .so> [workloads...]
# (the in-tree library is "new", the given one "base"; alternating runs, 60 steps each)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
other=$1; shift
wls=${@:-target_10M_400x200_4pass config3_1M_400x200_4pass config2_100k_400x200_1pass}
for w in $wls; do for v in new base new base; do
  if [ $v = base ]; then export RADARAYS_MI355_LIB=$other; else unset RADARAYS_MI355_LIB; fi
  python3 $R/bench.py --workload $w --steps 60 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; print('$w $v', d['value'], 'alone', r['isolated']['avg_launch_us'], 'live', r['avg_launch_us'])
"; done; done
