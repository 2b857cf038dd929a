#!/bin/bash
# round 6: the GPU suite on the final library, then the target's profiles once more (counter read-backs now by kernel stores)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd $R; mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
bash tools/r6_suite.sh
bash profiles/collect.sh r06 > gpurun_out/collect_r06.log 2>&1
grep -c rocclr gpurun_out/profiles_r06/kernel_stats.csv; grep rocclr gpurun_out/profiles_r06/kernel_stats.csv | cut -c1-80
bash profiles/collect.sh r06_c2 --workload config2_100k_400x200_1pass > gpurun_out/collect_r06_c2.log 2>&1
grep rocclr gpurun_out/profiles_r06_c2/kernel_stats.csv | cut -c1-80
