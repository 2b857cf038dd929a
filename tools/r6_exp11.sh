#!/bin/bash
# round 6, GPU call 11: run-to-run spread of config 2 through the SDMA route; active vs blocked wait of the workers
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd $R; mkdir -p gpurun_out; O=gpurun_out/r6_exp11.txt; : > $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
line() { python - "$1" <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); p=d["images_per_s_per_step"]
print("value", d["value"], "cadence median/p10/p90", p["median"], p["p10"], p["p90"], "route", d["config"].get("host_delivery"))
PY
}
nproc >> $O; cat /sys/fs/cgroup/cpu.max >> $O 2>/dev/null
C2="--workload config2_100k_400x200_1pass --no-cpu-baseline --no-extras --steps 100 --warmup 5"
for rep in 1 2 3 4; do for a in 600 0 2000; do
  RR_SDMA_ACTIVE_US=$a timeout 300 python bench.py $C2 > gpurun_out/r6_exp11_c2.json 2>/dev/null; echo "c2 active_us=$a: $(line gpurun_out/r6_exp11_c2.json)" >> $O
done; done
for rep in 1 2; do for a in 600 0; do
  RR_SDMA_ACTIVE_US=$a timeout 300 python bench.py --no-cpu-baseline --no-extras --steps 60 --warmup 5 > gpurun_out/r6_exp11_t.json 2>/dev/null; echo "target active_us=$a: $(line gpurun_out/r6_exp11_t.json)" >> $O
done; done
cat $O
