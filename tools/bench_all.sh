#!/bin/bash
# GPU box helper: the bench line of every workload (with CPU baseline) -> gpurun_out/<tag>_bench_<workload>.json
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd $R; mkdir -p gpurun_out; tag=${1:-r02}
for w in config2_100k_400x200_1pass config3_1M_400x200_4pass target_10M_400x200_4pass config4_10M_400x1000_4pass; do
  timeout 900 python bench.py --workload $w > gpurun_out/${tag}_bench_$w.json 2> gpurun_out/${tag}_bench_$w.err
  echo "$w $(grep -o '"value": [0-9.]*' gpurun_out/${tag}_bench_$w.json | head -2 | tr '\n' ' ')"
  timeout 600 python bench.py --workload $w --frames-per-rank 1 --no-cpu-baseline > gpurun_out/${tag}_bench_${w}_fpr1.json 2>/dev/null
  echo "  fpr1 $(grep -o '"value": [0-9.]*' gpurun_out/${tag}_bench_${w}_fpr1.json | head -1)"
done
w=config5_10M_400x1000_8pass_pertri
timeout 900 python bench.py --workload $w --frames-per-rank 1 --steps 40 --warmup 4 > gpurun_out/${tag}_bench_$w.json 2> gpurun_out/${tag}_bench_$w.err
echo "$w $(grep -o '"value": [0-9.]*' gpurun_out/${tag}_bench_$w.json | head -2 | tr '\n' ' ')"; tail -2 gpurun_out/${tag}_bench_$w.err
