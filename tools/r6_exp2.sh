#!/bin/bash
# round 6, GPU call 2: XCD-confined copy kernel on config 2; end-of-run flush on the target; parity of the root-preload build
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd $R; mkdir -p gpurun_out; O=gpurun_out/r6_exp2.txt; : > $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
pick() { grep -o "\"value\": [0-9.]*\|\"avg_launch_us\": [0-9.]*" | head -3 | tr '\n' ' '; }
run() { local label=$1; shift; local envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  echo "$label: $(env "${envs[@]}" timeout 300 python "$@" 2> gpurun_out/r6_exp2_err.log | pick)" >> $O; }
RADARAYS_MI355_LIB=$R/ab_libs/libradarays_rootpre.so timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/r6_exp2_rootpre_parity.log 2>&1
echo "rootpre parity rc=$? $(tail -1 gpurun_out/r6_exp2_rootpre_parity.log)" >> $O
timeout 600 python -m pytest tests/test_gpu_round6.py tests/test_gpu_round3.py -x -q -m gpu > gpurun_out/r6_exp2_pytest.log 2>&1; echo "pytest r6+r3 rc=$? $(tail -1 gpurun_out/r6_exp2_pytest.log)" >> $O
C2="--workload config2_100k_400x200_1pass --no-cpu-baseline --no-extras --steps 100 --warmup 5"
export RR_BENCH_LIVE_TIMING=0
for rep in 1 2; do
run "c2 memcpy" RR_FLUSH_KERNEL=0 -- bench.py $C2
for x in 0 3; do for b in 2 4 8 16 32; do for i in 0 2; do
  run "c2 xcd=$x blocks=$b inflight=$i" RR_FLUSH_XCD=$x RR_FLUSH_BLOCKS=$b RR_FLUSH_INFLIGHT=$i -- bench.py $C2
done; done; done
run "c2 xcd=-1 blocks=32 inflight=4" RR_FLUSH_XCD=-1 -- bench.py $C2
done
unset RR_BENCH_LIVE_TIMING
T="--no-cpu-baseline --no-extras --warmup 5"
for i in 1 2 3; do
  run "target60 r5-tree" X=1 -- ab_old/bench.py $T --steps 60
  run "target60 new flush=kernel" X=1 -- bench.py $T --steps 60
  run "target60 new flush=memcpy" RR_FLUSH_KERNEL=0 -- bench.py $T --steps 60
  run "target20 r5-tree" X=1 -- ab_old/bench.py $T --steps 20
  run "target20 new flush=kernel" X=1 -- bench.py $T --steps 20
  run "target20 new flush=kernel xcd=-1 b=64 i=0" RR_FLUSH_XCD=-1 RR_FLUSH_BLOCKS=64 RR_FLUSH_INFLIGHT=0 -- bench.py $T --steps 20
  run "target20 new flush=memcpy" RR_FLUSH_KERNEL=0 -- bench.py $T --steps 20
done
cat $O
