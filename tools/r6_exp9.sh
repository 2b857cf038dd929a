#!/bin/bash
# round 6, GPU call 9: two SDMA workers; what the launch-graph guard costs; does an unguarded replay show stale poses?
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd $R; mkdir -p gpurun_out; O=gpurun_out/r6_exp9.txt; : > $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_gpu_round6.py tests/test_gpu_round3.py tests/test_gpu_round5.py tests/test_gpu_multi.py -x -q -m gpu > gpurun_out/r6_exp9_pytest.log 2>&1; echo "pytest rc=$? $(tail -1 gpurun_out/r6_exp9_pytest.log)" >> $O
for i in 1 2 3; do RR_GRAPH_GUARD=0 timeout 300 python -m pytest tests/test_gpu_round6.py -x -q -m gpu -k back_to_back > gpurun_out/r6_exp9_noguard.log 2>&1; echo "unguarded replays, back-to-back test run $i: rc=$? $(tail -1 gpurun_out/r6_exp9_noguard.log)" >> $O; done
line() { python - "$1" <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
print("value", d["value"], "hbm", (d.get("hbm_resident") or {}).get("value"), "single_pose", (d.get("single_pose") or {}).get("value"), "sync_ms", (d.get("single_frame_sync") or {}).get("ms_per_frame"), "graphs", d.get("launch_graphs", {}).get("replays_in_timed_region"), "route", d["config"].get("host_delivery"))
PY
}
for g in 1 0; do for rep in 1 2; do
  RR_GRAPH_GUARD=$g timeout 400 python bench.py --no-cpu-baseline --workload config2_100k_400x200_1pass > gpurun_out/r6_exp9_c2_g$g.json 2> gpurun_out/r6_exp9_err.log; echo "c2 guard=$g: $(line gpurun_out/r6_exp9_c2_g$g.json)" >> $O
  RR_GRAPH_GUARD=$g timeout 400 python bench.py --no-cpu-baseline --workload config2_100k_400x200_1pass --frames-per-rank 1 > gpurun_out/r6_exp9_c2f1_g$g.json 2>> gpurun_out/r6_exp9_err.log; echo "c2 fpr1 guard=$g: $(line gpurun_out/r6_exp9_c2f1_g$g.json)" >> $O
done; done
timeout 400 python bench.py --no-cpu-baseline > gpurun_out/r6_exp9_target.json 2>> gpurun_out/r6_exp9_err.log; echo "target: $(line gpurun_out/r6_exp9_target.json)" >> $O
for cfg in 2 4; do echo "== C++ caller config $cfg SDMA route (two workers)" >> $O; bash tools/cpp_bench.sh 8000 8 multi $cfg 2>&1 | tail -3 >> $O; done
cat $O
