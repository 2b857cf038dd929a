#!/bin/bash
# Runs ON THE GPU BOX: extra PMC passes (TCP / UTCL1 / busy counters) of the bench command -> gpurun_out/pmc_extra/summary.json
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmc_extra; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
ARGS="--steps 30 --warmup 5 --no-cpu-baseline --no-extras"
i=0
# (a set of TA_* counters made rocprofv3 abort with signal 6 on this pool: left out)
for set in "TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum" "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_TCC_READ_REQ_sum" "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES TCP_TOTAL_ACCESSES_sum TD_TD_BUSY_sum"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/p$i -- python3 $R/bench.py $ARGS > /dev/null 2> $OUT/p$i.log
done
python3 - <<PY
import csv, glob, collections, json
out = {}
for d in sorted(glob.glob("$OUT/p*")):
    f = glob.glob(d + "/*/*counter_collection.csv")
    if not f: continue
    acc = collections.defaultdict(float); n = collections.Counter()
    for row in csv.DictReader(open(f[0])):
        k = (row["Kernel_Name"].split("(")[0], row["Counter_Name"])
        acc[k] += float(row["Counter_Value"]); n[k] += 1
    for (k, c), v in acc.items():
        if "rr::" in k and "true, false" not in k.replace("<true, false, false, false>", ""): out.setdefault(k, {})[c] = round(v / n[(k, c)], 1)
json.dump(out, open("$OUT/summary.json", "w"), indent=1, sort_keys=True)
for k, v in out.items():
    if "k_trace" in k: print(k, v)
PY
rm -rf $OUT/p1 $OUT/p2 $OUT/p3
