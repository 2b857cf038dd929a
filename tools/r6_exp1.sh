#!/bin/bash
# round 6, GPU call 1: new GPU tests; config-2 host-delivery bisect + copy-kernel sweep; target A/B (r5 tree / base / root preload)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd $R; mkdir -p gpurun_out; O=gpurun_out/r6_exp1.txt; : > $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
pick() { grep -o "\"value\": [0-9.]*\|\"avg_launch_us\": [0-9.]*" | tr '\n' ' '; }
timeout 900 python -m pytest tests/test_gpu_round6.py -x -q > gpurun_out/r6_exp1_pytest.log 2>&1; echo "pytest round6 rc=$? $(tail -1 gpurun_out/r6_exp1_pytest.log)" >> $O
C2="--workload config2_100k_400x200_1pass --no-cpu-baseline --no-extras --steps 100 --warmup 5"
run() { # label, env..., then -- bench args
  local label=$1; shift
  local envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  echo "$label: $(env "${envs[@]}" timeout 300 python "$@" 2> gpurun_out/r6_exp1_err.log | pick)" >> $O
}
run "c2 r5-tree" X=1 -- ab_old/bench.py $C2
run "c2 new flush=memcpy" RR_FLUSH_KERNEL=0 -- bench.py $C2
run "c2 new flush=memcpy no-live-timing" RR_FLUSH_KERNEL=0 RR_BENCH_LIVE_TIMING=0 -- bench.py $C2
run "c2 new default(32,4)" X=1 -- bench.py $C2
run "c2 new default no-live-timing" RR_BENCH_LIVE_TIMING=0 -- bench.py $C2
for bi in "8 0" "16 0" "32 0" "64 0" "128 0" "16 4" "64 1" "64 4" "128 2" "256 1"; do set -- $bi
  run "c2 new blocks=$1 inflight=$2" RR_FLUSH_BLOCKS=$1 RR_FLUSH_INFLIGHT=$2 -- bench.py $C2
done
run "c2 r5-tree again" X=1 -- ab_old/bench.py $C2
T="--no-cpu-baseline --no-extras --steps 60 --warmup 5"
for i in 1 2; do
  run "target r5-tree" X=1 -- ab_old/bench.py $T
  run "target base" RADARAYS_MI355_LIB=$R/ab_libs/libradarays_base.so -- bench.py $T
  run "target rootpre" RADARAYS_MI355_LIB=$R/ab_libs/libradarays_rootpre.so -- bench.py $T
done
cat $O
