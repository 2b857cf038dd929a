#!/bin/bash
# round 6, GPU call 7: host delivery over SDMA through ROCr (rr_sdma.cpp)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd $R; mkdir -p gpurun_out; O=gpurun_out/r6_exp7.txt; : > $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
pick() { grep -o "\"value\": [0-9.]*\|\"avg_launch_us\": [0-9.]*" | head -3 | tr '\n' ' '; }
run() { local label=$1; shift; local envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  echo "$label: $(env "${envs[@]}" timeout 300 python "$@" 2> gpurun_out/r6_exp7_err.log | pick) $(grep -h '\[rr\]' gpurun_out/r6_exp7_err.log | head -2 | tr '\n' ' ')" >> $O; }
export RR_HOST_SDMA_VERBOSE=1
timeout 900 python -m pytest tests/test_gpu_round6.py tests/test_gpu_round3.py tests/test_gpu_round5.py tests/test_gpu_fullsize.py -x -q -m gpu > gpurun_out/r6_exp7_pytest.log 2>&1; echo "pytest rc=$? $(tail -1 gpurun_out/r6_exp7_pytest.log) $(grep -c '\[rr\]' gpurun_out/r6_exp7_pytest.log) [rr] lines" >> $O
C2="--workload config2_100k_400x200_1pass --no-cpu-baseline --no-extras --steps 100 --warmup 5"
T="--no-cpu-baseline --no-extras --warmup 5"
for rep in 1 2 3; do
run "c2 SDMA" X=1 -- bench.py $C2
run "c2 no-sdma memcpy" RR_HOST_SDMA=0 RR_FLUSH_KERNEL=0 -- bench.py $C2
run "target60 SDMA" X=1 -- bench.py $T --steps 60
run "target60 no-sdma (trickle + flush kernel)" RR_HOST_SDMA=0 -- bench.py $T --steps 60
run "target20 SDMA" X=1 -- bench.py $T --steps 20
run "target20 no-sdma" RR_HOST_SDMA=0 -- bench.py $T --steps 20
done
run "config3 SDMA" X=1 -- bench.py $T --steps 60 --workload config3_1M_400x200_4pass
run "config3 no-sdma" RR_HOST_SDMA=0 -- bench.py $T --steps 60 --workload config3_1M_400x200_4pass
cat $O
