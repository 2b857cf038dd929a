#!/bin/bash
# round 6, GPU call 8: SDMA route with two image buffers per lane (the host no longer waits for the lane's previous copy)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd $R; mkdir -p gpurun_out; O=gpurun_out/r6_exp8.txt; : > $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
pick() { grep -o "\"value\": [0-9.]*" | head -2 | tr '\n' ' '; }
run() { local label=$1; shift; local envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  echo "$label: $(env "${envs[@]}" timeout 400 python "$@" 2> gpurun_out/r6_exp8_err.log | pick)" >> $O; }
RR_HOST_SDMA_VERBOSE=1 timeout 900 python -m pytest tests/test_gpu_round6.py tests/test_gpu_round3.py tests/test_gpu_round5.py tests/test_gpu_fullsize.py tests/test_gpu_multi.py -x -q -m gpu > gpurun_out/r6_exp8_pytest.log 2>&1; echo "pytest rc=$? $(tail -1 gpurun_out/r6_exp8_pytest.log)" >> $O
T="--no-cpu-baseline --no-extras --warmup 5"
for rep in 1 2; do
run "c2" X=1 -- bench.py $T --steps 100 --workload config2_100k_400x200_1pass
run "c2 fpr1" X=1 -- bench.py $T --steps 100 --workload config2_100k_400x200_1pass --frames-per-rank 1
run "target60" X=1 -- bench.py $T --steps 60
run "target20" X=1 -- bench.py $T --steps 20
run "target fpr1" X=1 -- bench.py $T --steps 60 --frames-per-rank 1
run "target fpr1 no-sdma" RR_HOST_SDMA=0 -- bench.py $T --steps 60 --frames-per-rank 1
run "config5" X=1 -- bench.py $T --workload config5_10M_400x1000_8pass_pertri --frames-per-rank 1 --steps 40 --warmup 4
run "config5 no-sdma" RR_HOST_SDMA=0 -- bench.py $T --workload config5_10M_400x1000_8pass_pertri --frames-per-rank 1 --steps 40 --warmup 4
done
for cfg in 2 4; do
  echo "== C++ caller config $cfg SDMA route" >> $O; bash tools/cpp_bench.sh 8000 8 multi $cfg 2>&1 | tail -3 >> $O
done
echo "== C++ caller config 2, RR_HOST_SDMA=0 RR_FLUSH_KERNEL=0 (the runtime's own engine choice)" >> $O; RR_HOST_SDMA=0 RR_FLUSH_KERNEL=0 bash tools/cpp_bench.sh 8000 8 multi 2 2>&1 | tail -3 >> $O
cat $O
