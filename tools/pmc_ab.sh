cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in new base; do
  if [ $v = base ]; then export RADARAYS_MI355_LIB=$R/build/base_lib.so; else unset RADARAYS_MI355_LIB; fi
  export RR_GRAPHS=0
  rm -rf /tmp/pmc_$v
  timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d /tmp/pmc_$v -- python3 $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-extras --slots 1 > /dev/null 2> /tmp/pmc_$v.log
  python3 - <<PY
import csv, glob, collections
f = glob.glob("/tmp/pmc_$v/*/*counter_collection.csv")
acc = collections.defaultdict(float); n = collections.Counter()
for row in csv.DictReader(open(f[0])):
    if "k_trace<false, false, false, true>" not in row["Kernel_Name"]: continue
    acc[row["Counter_Name"]] += float(row["Counter_Value"]); n[row["Counter_Name"]] += 1
print("$v", {k: round(acc[k] / n[k] / 1e6, 2) for k in sorted(acc)}, "launches", max(n.values()) if n else 0)
PY
done
