#!/bin/bash
# round 6, GPU call 6: one-pass frames copied out on a dedicated stream (kernel / memcpy); contention of stack vs stackless
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd $R; mkdir -p gpurun_out; O=gpurun_out/r6_exp6.txt; : > $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
pick() { grep -o "\"value\": [0-9.]*" | head -1 | tr '\n' ' '; }
run() { local label=$1; shift; local envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  echo "$label: $(env "${envs[@]}" timeout 300 python "$@" 2> gpurun_out/r6_exp6_err.log | pick)" >> $O; }
timeout 600 python -m pytest tests/test_gpu_round6.py tests/test_gpu_round3.py -x -q -m gpu > gpurun_out/r6_exp6_pytest.log 2>&1; echo "pytest rc=$? $(tail -1 gpurun_out/r6_exp6_pytest.log)" >> $O
RR_HOST_COPY_STREAM=1 timeout 600 python -m pytest tests/test_gpu_round6.py tests/test_gpu_round3.py -x -q -m gpu > gpurun_out/r6_exp6_pytest2.log 2>&1; echo "pytest (copy stream) rc=$? $(tail -1 gpurun_out/r6_exp6_pytest2.log)" >> $O
C2="--workload config2_100k_400x200_1pass --no-cpu-baseline --no-extras --steps 100 --warmup 5"
export RR_BENCH_LIVE_TIMING=0
for rep in 1 2; do
run "c2 memcpy deferred" RR_FLUSH_KERNEL=0 -- bench.py $C2
run "c2 copy-stream memcpy" RR_HOST_COPY_STREAM=1 RR_FLUSH_KERNEL=0 -- bench.py $C2
run "c2 copy-stream memcpy slots=3" RR_HOST_COPY_STREAM=1 RR_FLUSH_KERNEL=0 -- bench.py $C2 --slots 3
for b in 8 16 32 64; do for x in -1 3; do
run "c2 copy-stream kernel blocks=$b xcd=$x" RR_HOST_COPY_STREAM=1 RR_FLUSH_BLOCKS=$b RR_FLUSH_THREADS=256 RR_FLUSH_INFLIGHT=0 RR_FLUSH_XCD=$x -- bench.py $C2
done; done
run "c2 copy-stream kernel blocks=16 slots=3" RR_HOST_COPY_STREAM=1 RR_FLUSH_BLOCKS=16 RR_FLUSH_THREADS=256 RR_FLUSH_INFLIGHT=0 RR_FLUSH_XCD=-1 -- bench.py $C2 --slots 3
run "c2 copy-stream kernel blocks=16 GPU_MAX_HW_QUEUES=8" GPU_MAX_HW_QUEUES=8 RR_HOST_COPY_STREAM=1 RR_FLUSH_BLOCKS=16 RR_FLUSH_THREADS=256 RR_FLUSH_INFLIGHT=0 RR_FLUSH_XCD=-1 -- bench.py $C2
run "c2 copy-stream memcpy GPU_MAX_HW_QUEUES=8" GPU_MAX_HW_QUEUES=8 RR_HOST_COPY_STREAM=1 RR_FLUSH_KERNEL=0 -- bench.py $C2
done
unset RR_BENCH_LIVE_TIMING
T="--no-cpu-baseline --no-extras --warmup 5 --steps 60"
timeout 300 python bench.py $T > gpurun_out/r6_exp6_target_stack.json 2>> gpurun_out/r6_exp6_err.log
RR_STACKLESS=1 timeout 300 python bench.py $T > gpurun_out/r6_exp6_target_stackless.json 2>> gpurun_out/r6_exp6_err.log
python - <<'PY' >> $O
import json
for n in ("stack", "stackless"):
    d = json.load(open("gpurun_out/r6_exp6_target_%s.json" % n))
    c = d["contention"]
    print("target", n, "value", d["value"], "live_us", c["live_us"], "alone_us", c["alone_us"], "isolated_sum", c["batch_ms_isolated_sum"], "live", c["batch_ms_live"])
PY
cat $O
