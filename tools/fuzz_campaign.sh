#!/bin/bash
# GPU box helper: the round's fuzz / soak campaign against the in-tree library; log -> gpurun_out/<tag>_fuzz.log
# usage: tools/fuzz_campaign.sh <tag> [scale = 1]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd $R; mkdir -p gpurun_out
tag=${1:-r06}; k=${2:-1}; L=gpurun_out/${tag}_fuzz.log
python - <<PY > $L
import sys; sys.path.insert(0, "$R")
import bench; print("kernel sources", bench.kernel_source_hash())
PY
run() { echo "== $*" >> $L; local t0=$(date +%s); timeout 1500 "$@" 2>&1 | grep -v amdgpu.ids | tail -4 >> $L; echo "   ($(( $(date +%s) - t0 )) s)" >> $L; }
run python tests/fuzz/fuzz_trace.py $((600 * k)) 9000
run python tests/fuzz/fuzz_diff2.py $((2500 * k)) 55
run python tests/fuzz/fuzz_batch.py $((500 * k)) 55
run python tests/fuzz/fuzz_state.py $((1200 * k)) 55
run python tests/fuzz/fuzz_sets.py $((800 * k)) 5
run python tests/fuzz/fuzz_big.py $((30 * k)) 55
RR_FUZZ_SEEDS=50000:$((50000 + 3000 * k)) run python -m pytest tests/test_gpu_parity.py -q -k random_differential
run python tools/soak.py $((6000 * k)) 4
run python tools/soak_host.py $((600 * k)) 4
run python tools/soak_multi.py $((400 * k)) 8 2
run python tools/soak_multi.py $((120 * k)) 8 3
run python tools/soak_multi.py $((150 * k)) 3 3
RR_MULTI_THREADS=0 run python tools/soak_multi.py $((200 * k)) 5 2
RR_TIGHT_FORCE=2 run python tests/fuzz/fuzz_batch.py $((200 * k)) 91
RR_GRAPHS=0 run python tests/fuzz/fuzz_state.py $((300 * k)) 91
# round 6: the routes the host delivery can take, launch graphs replayed by ONE lane with changing poses, the stack-free traversal
run python tests/fuzz/fuzz_trace.py $((300 * k)) 100000      # spatially split faces at grazing incidence
RR_HOST_SDMA=0 run python tools/soak_host.py $((300 * k)) 4
RR_HOST_SDMA=0 RR_FLUSH_KERNEL=0 run python tools/soak_host.py $((200 * k)) 2
run python tools/soak_host.py $((600 * k)) 2
RR_LANES=1 run python tools/soak.py $((1500 * k)) 4
RR_STACKLESS=1 run python tests/fuzz/fuzz_diff2.py $((600 * k)) 77
RR_STACKLESS=1 run python tests/fuzz/fuzz_batch.py $((150 * k)) 77
cat $L
