#!/bin/bash
# GPU box helper: the round's bench lines (every workload + the driver's shapes) -> gpurun_out/<tag>_bench_*.json
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd $R; tag=${1:-r05}
tools/bench_all.sh $tag 2>&1 | tail -12
python bench.py --steps 20 --warmup 5 > gpurun_out/${tag}_bench_target_20.json 2>/dev/null
python bench.py --no-cpu-baseline --self-launch > gpurun_out/${tag}_bench_target_selflaunch.json 2>/dev/null
python bench.py --no-cpu-baseline --force-slots > gpurun_out/${tag}_bench_target_forceslots.json 2>/dev/null
python bench.py --no-cpu-baseline --force-slots --strong > gpurun_out/${tag}_bench_target_forceslots_strong.json 2>/dev/null
for f in target_20 target_selflaunch target_forceslots target_forceslots_strong; do echo "$f $(grep -o '"value": [0-9.]*' gpurun_out/${tag}_bench_$f.json | head -1)"; done
