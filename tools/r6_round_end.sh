#!/bin/bash
# round 6, the evidence of the final build in ONE call: rocprofv3 kernel traces + PMC of every workload, counters re-keyed to the
# sources, the bench lines, the fuzz / soak campaign.  Everything that must come home is put under gpurun_out/.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd $R; mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
SKIP_BENCH=1 bash tools/final_collect.sh r06 > gpurun_out/r6_round_end_collect.log 2>&1
for t in "" _c2 _c3 _c4 _c5; do d=gpurun_out/profiles_r06$t; for f in kernel_stats.csv kernel_stats_isolated.csv pmc_summary.json bench_under_rocprof.json bench_under_rocprof_isolated.json copy_engine.txt; do cp $d/$f profiles/r06${t}_$f; done; done
python profiles/make_counters.py r06 target_10M_400x200_4pass= config3_1M_400x200_4pass=_c3 config2_100k_400x200_1pass=_c2 config4_10M_400x1000_4pass=_c4 config5_10M_400x1000_8pass_pertri=_c5 > gpurun_out/r6_make_counters.log 2>&1
cp profiles/roofline_counters.json gpurun_out/roofline_counters.json
bash tools/final_bench.sh r06 2>&1 | tail -16
bash tools/fuzz_campaign.sh r06 > /dev/null 2>&1; tail -60 gpurun_out/r06_fuzz.log
