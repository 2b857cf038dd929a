#!/bin/bash
# GPU box helper: frames-per-step sweep at the driver's (20/5) and the default (400/40) step counts
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd $R; mkdir -p gpurun_out
W=${1:-config2_100k_400x200_1pass}
for f in 1 2 4 8 16 32; do
  a=$(timeout 300 python bench.py --no-cpu-baseline --workload $W --frames-per-rank $f --steps 20 --warmup 5 2>/dev/null | grep -o "\"value\": [0-9.]*")
  b=$(timeout 300 python bench.py --no-cpu-baseline --workload $W --frames-per-rank $f --steps $((1600 / f)) --warmup $((160 / f + 4)) 2>/dev/null | grep -o "\"value\": [0-9.]*\|\"avg_launch_us\": [0-9.]*" | tr '\n' ' ')
  echo "fpr $f  driver-args(20/5): $a   long: $b"
done
