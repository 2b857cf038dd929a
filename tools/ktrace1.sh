#!/bin/bash
# GPU box helper: isolated kernel durations (one frame lane, one frame per step) under rocprofv3
# usage: tools/ktrace1.sh <workload> [extra bench args]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; W=${1:-config2_100k_400x200_1pass}; shift
export RR_LANES=1; cd /tmp; export TMPDIR=/tmp; rm -rf $R/gpurun_out/kt1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/kt1 -- python3 $R/bench.py --workload $W --frames-per-rank ${FPR:-1} --slots 1 --steps 60 --warmup 10 --no-cpu-baseline "$@" > $R/gpurun_out/kt1.log 2>&1
python3 - <<PY
import csv, glob
f = glob.glob("$R/gpurun_out/kt1/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:10]:
    print("%-58s calls %5s avg %9.1f us total %8.2f ms" % (r["Name"][:58], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
grep -o '"value": [0-9.]*' $R/gpurun_out/kt1.log; rm -rf $R/gpurun_out/kt1
