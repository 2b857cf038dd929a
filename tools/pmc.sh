#!/bin/bash
# usage: scripts_pmc.sh <tag> <probe args...>   -- PMC passes for the probe (no kernel-trace mixing other domains)
R=$GRAFT_REPO_ROOT; tag=$1; shift
cd /tmp; export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" \
           "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum GRBM_GUI_ACTIVE" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 150 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$tag/p$i -- python3 $R/tools/probe.py "$@" > /dev/null 2>&1
  echo "pass $i rc=$?"
done
python3 - <<PY
import csv, glob, collections
for d in sorted(glob.glob("$R/gpurun_out/pmc_$tag/p*")):
    f = glob.glob(d + "/*/*counter_collection.csv")
    if not f: print(d, "no counters"); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for row in csv.DictReader(open(f[0])):
        k = row["Kernel_Name"].split("(")[0][:40]
        acc[k][row["Counter_Name"]] += float(row["Counter_Value"]); 
    for row in csv.DictReader(open(f[0])):
        pass
    disp = collections.Counter()
    for row in csv.DictReader(open(f[0])):
        disp[(row["Kernel_Name"].split("(")[0][:40], row["Counter_Name"])] += 1
    for k in acc:
        if "rr::" not in k: continue
        print(k, {c: round(v / disp[(k, c)], 1) for c, v in acc[k].items()})
PY
