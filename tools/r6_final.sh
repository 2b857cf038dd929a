#!/bin/bash
# round 6: bench lines of every workload + the driver's shapes, the C++ caller, the N > 1 loop on one rank
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd $R; mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
bash tools/final_bench.sh r06 2>&1 | tail -20
for cfg in 2 4; do
  echo "== C++ caller (tools/cpp_bench.cpp multi), config $cfg, SDMA route"; bash tools/cpp_bench.sh 8000 8 multi $cfg 2>&1 | tail -4
  echo "== C++ caller, config $cfg, RR_HOST_SDMA=0"; RR_HOST_SDMA=0 bash tools/cpp_bench.sh 8000 8 multi $cfg 2>&1 | tail -4
done
