#!/bin/bash
# GPU box helper: SQ_ACTIVE_INST_VALU / SQ_INSTS_VALU per kernel (is any kernel paying more than one issue slot per VALU instruction?)
cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; export RR_GRAPHS=0
rm -rf /tmp/pmc_v
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_INSTS_SALU --kernel-trace --output-format csv -d /tmp/pmc_v -- python3 $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-extras --slots 1 "$@" > /dev/null 2> /tmp/pmc_v.log
python3 - <<PY
import csv, glob, collections
f = glob.glob("/tmp/pmc_v/*/*counter_collection.csv")
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for row in csv.DictReader(open(f[0])):
    k = row["Kernel_Name"].split("(")[0].replace("void rr::", "")
    acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
for k, v in sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_INSTS_VALU", 0)):
    if v.get("SQ_INSTS_VALU", 0) > 1e6:
        print("%-48s insts %9.1f M  active/insts %.3f  salu/valu %.2f" % (k[:48], v["SQ_INSTS_VALU"] / 1e6, v["SQ_ACTIVE_INST_VALU"] / v["SQ_INSTS_VALU"], v["SQ_INSTS_SALU"] / v["SQ_INSTS_VALU"]))
PY
