#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): rocprofv3 kernel-trace stats and PMC passes of the SAME command the
# driver benches (python bench.py; default workload = the north-star target), summaries under gpurun_out/.
# usage: profiles/collect.sh <round-tag> [bench args...]        e.g.  collect.sh r03_c3 --workload config3_1M_400x200_4pass
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
tag=${1:-r06}; shift
OUT=$R/gpurun_out/profiles_$tag; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
ARGS="--steps 60 --warmup 5 --no-cpu-baseline --no-extras $@"
export RR_GRAPHS=0   # every chain kernel by kernel (as in rounds 3-5, whose timed loop carried host copies and was never replayed from a graph)
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ktrace -- python3 $R/bench.py $ARGS > $OUT/bench_under_rocprof.json 2> $OUT/ktrace.log
cp $(find $OUT/ktrace -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv 2>/dev/null
# the same launches with ONE batch on the GPU at a time (one stream, one buffer set): what roofline.isolated divides by
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ktrace1 -- python3 $R/bench.py $ARGS --slots 1 > $OUT/bench_under_rocprof_isolated.json 2> $OUT/ktrace1.log
cp $(find $OUT/ktrace1 -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats_isolated.csv 2>/dev/null
# (round 6) who carries the images to host memory: the kernel trace again with the memory-copy domain -- the SDMA jobs of
# csrc/rr_sdma.cpp show up as MEMORY_COPY_DEVICE_TO_HOST records, a runtime's blit copies as __amd_rocclr_copyBuffer kernels
timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d $OUT/ktrace2 -- python3 $R/bench.py $ARGS > /dev/null 2> $OUT/ktrace2.log
for f in $(find $OUT/ktrace2 -name "*_stats.csv"); do echo "== $(basename $f)"; head -6 $f | cut -c1-160; done > $OUT/copy_engine.txt 2>/dev/null
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_$c -- python3 $R/bench.py $ARGS > /dev/null 2> $OUT/pmc_$c.log
done
timeout 600 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum --kernel-trace --output-format csv -d $OUT/pmc_cache -- python3 $R/bench.py $ARGS > /dev/null 2> $OUT/pmc_cache.log
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $OUT/pmc_sq -- python3 $R/bench.py $ARGS > /dev/null 2> $OUT/pmc_sq.log
timeout 600 rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $OUT/pmc_lds -- python3 $R/bench.py $ARGS > /dev/null 2> $OUT/pmc_lds.log
python3 - <<PY
import csv, glob, collections, json
out = {}
for d in sorted(glob.glob("$OUT/pmc_*")):
    f = glob.glob(d + "/*/*counter_collection.csv")
    if not f: continue
    acc = collections.defaultdict(float); n = collections.Counter()
    for row in csv.DictReader(open(f[0])):
        k = (row["Kernel_Name"].split("(")[0], row["Counter_Name"])
        acc[k] += float(row["Counter_Value"]); n[k] += 1
    for (k, c), v in acc.items():
        if "rr::" in k: out.setdefault(k, {})[c] = {"mean_per_launch": v / n[(k, c)], "launches": n[(k, c)]}
json.dump(out, open("$OUT/pmc_summary.json", "w"), indent=1, sort_keys=True)
print(json.dumps({k: {c: round(x["mean_per_launch"], 1) for c, x in v.items()} for k, v in out.items()}, indent=1))
PY
rm -rf $OUT/ktrace $OUT/ktrace1 $OUT/ktrace2 $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE $OUT/pmc_cache $OUT/pmc_sq $OUT/pmc_lds
ls -la $OUT
