#!/bin/bash
# round 6, GPU call 5: unrolled copy kernel on config 2; the stack-free traversal (tests + target / config 3 / config 2)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd $R; mkdir -p gpurun_out; O=gpurun_out/r6_exp5.txt; : > $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
pick() { grep -o "\"value\": [0-9.]*\|\"avg_launch_us\": [0-9.]*" | head -3 | tr '\n' ' '; }
run() { local label=$1; shift; local envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  echo "$label: $(env "${envs[@]}" timeout 300 python "$@" 2> gpurun_out/r6_exp5_err.log | pick)" >> $O; }
RR_STACKLESS=1 timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round6.py -x -q -m gpu > gpurun_out/r6_exp5_sl_pytest.log 2>&1; echo "stackless parity+r6 rc=$? $(tail -1 gpurun_out/r6_exp5_sl_pytest.log)" >> $O
RR_STACKLESS=1 timeout 600 python tests/fuzz/fuzz_diff2.py 60 > gpurun_out/r6_exp5_sl_fuzz.log 2>&1; echo "stackless fuzz_diff2 rc=$? $(tail -1 gpurun_out/r6_exp5_sl_fuzz.log)" >> $O
T="--no-cpu-baseline --no-extras --warmup 5 --steps 60"
for i in 1 2; do
  run "target stack" X=1 -- bench.py $T
  run "target STACKLESS" RR_STACKLESS=1 -- bench.py $T
done
run "config3 stack" X=1 -- bench.py $T --workload config3_1M_400x200_4pass
run "config3 STACKLESS" RR_STACKLESS=1 -- bench.py $T --workload config3_1M_400x200_4pass
C2="--workload config2_100k_400x200_1pass --no-cpu-baseline --no-extras --steps 100 --warmup 5"
export RR_BENCH_LIVE_TIMING=0
run "c2 stack (memcpy)" RR_FLUSH_KERNEL=0 -- bench.py $C2
run "c2 STACKLESS (memcpy)" RR_FLUSH_KERNEL=0 RR_STACKLESS=1 -- bench.py $C2
for u in 2 4 8; do for t in 256; do for b in 4 8 16 32 64; do for i in 0 8; do
  run "c2 unroll=$u threads=$t blocks=$b inflight=$i xcd=-1" RR_FLUSH_UNROLL=$u RR_FLUSH_XCD=-1 RR_FLUSH_THREADS=$t RR_FLUSH_BLOCKS=$b RR_FLUSH_INFLIGHT=$i -- bench.py $C2
done; done; done; done
for u in 4 8; do for b in 4 8 16; do
  run "c2 unroll=$u threads=256 blocks=$b inflight=0 xcd=3" RR_FLUSH_UNROLL=$u RR_FLUSH_XCD=3 RR_FLUSH_THREADS=256 RR_FLUSH_BLOCKS=$b RR_FLUSH_INFLIGHT=0 -- bench.py $C2
done; done
run "c2 memcpy" RR_FLUSH_KERNEL=0 -- bench.py $C2
cat $O
