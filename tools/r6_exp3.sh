#!/bin/bash
# round 6, GPU call 3: shape of the copy kernel (fat workgroups = few CUs; nontemporal stores) on config 2
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd $R; mkdir -p gpurun_out; O=gpurun_out/r6_exp3.txt; : > $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
pick() { grep -o "\"value\": [0-9.]*" | head -1 | tr '\n' ' '; }
run() { local label=$1; shift; local envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  echo "$label: $(env "${envs[@]}" timeout 300 python "$@" 2> gpurun_out/r6_exp3_err.log | pick)" >> $O; }
timeout 600 python -m pytest tests/test_gpu_round6.py -x -q -m gpu -s > gpurun_out/r6_exp3_pytest.log 2>&1; echo "pytest r6 rc=$? $(tail -1 gpurun_out/r6_exp3_pytest.log)" >> $O
grep "fresnel GPU vs oracle" gpurun_out/r6_exp3_pytest.log >> $O
C2="--workload config2_100k_400x200_1pass --no-cpu-baseline --no-extras --steps 100 --warmup 5"
export RR_BENCH_LIVE_TIMING=0
run "c2 memcpy" RR_FLUSH_KERNEL=0 -- bench.py $C2
for x in -1 3; do for t in 256 1024; do for b in 1 2 4 8 16; do for i in 0 2; do
  run "c2 xcd=$x threads=$t blocks=$b inflight=$i" RR_FLUSH_XCD=$x RR_FLUSH_THREADS=$t RR_FLUSH_BLOCKS=$b RR_FLUSH_INFLIGHT=$i -- bench.py $C2
done; done; done; done
for x in -1 3; do for t in 64 1024; do for b in 4 16; do
  run "c2 NT xcd=$x threads=$t blocks=$b inflight=0" RR_FLUSH_NT=1 RR_FLUSH_XCD=$x RR_FLUSH_THREADS=$t RR_FLUSH_BLOCKS=$b RR_FLUSH_INFLIGHT=0 -- bench.py $C2
done; done; done
run "c2 memcpy" RR_FLUSH_KERNEL=0 -- bench.py $C2
cat $O
