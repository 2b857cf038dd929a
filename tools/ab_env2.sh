#!/bin/bash
# GPU box helper: A/B over pairs "A:B" of RR_BEAM_SORT:RR_BEAM_SORT2
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd $R; mkdir -p gpurun_out
pairs=$1; shift
W=${@:-config2_100k_400x200_1pass config3_1M_400x200_4pass target_10M_400x200_4pass}
for pr in $pairs; do
  export RR_BEAM_SORT=${pr%%:*} RR_BEAM_SORT2=${pr##*:}
  for w in $W; do
    timeout 300 python bench.py --no-cpu-baseline --workload $w > gpurun_out/abenv2_$w.log 2>&1
    echo "sort=$pr $w $(grep -o "\"value\": [0-9.]*\|\"avg_launch_us\": [0-9.]*" gpurun_out/abenv2_$w.log | tr '\n' ' ')"
  done
done
