#!/bin/bash
# round 6, final build (set-up uploads and counter read-backs by the library's own kernels): GPU suite, then everything r6_round_end.sh does
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd $R; mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
bash tools/r6_suite.sh
bash tools/r6_round_end.sh > gpurun_out/r6_round_end2.log 2>&1
for t in "" _c2 _c3 _c4 _c5; do echo "rocclr in r06$t: $(grep -c rocclr_copy gpurun_out/profiles_r06$t/kernel_stats.csv) copy, $(grep -c rocclr_fill gpurun_out/profiles_r06$t/kernel_stats.csv) fill"; done
tail -45 gpurun_out/r6_round_end2.log | cut -c1-200
