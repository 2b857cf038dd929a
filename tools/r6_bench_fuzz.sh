#!/bin/bash
# round 6: the bench lines of the final library + the fuzz / soak campaign
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd $R; mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
bash tools/final_bench.sh r06 2>&1 | tail -16
bash tools/fuzz_campaign.sh r06 > /dev/null 2>&1; grep -v "^\.\.\.\.\.\." gpurun_out/r06_fuzz.log | tail -80
