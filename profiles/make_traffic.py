#!/usr/bin/env python3
"""profiles/roofline_traffic.json from the PMC summaries that profiles/collect.sh writes:
HBM bytes per k_trace launch = 2 x FETCH_SIZE + WRITE_SIZE (KB -> B), launch-weighted mean over the
k_trace variants of all passes (the instrumented stats launches are left out).
usage: profiles/make_traffic.py <round tag, e.g. r02b>"""
import json, os, sys
here = os.path.dirname(os.path.abspath(__file__))
tag = sys.argv[1]
out = {"_how": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, with --kernel-trace only) of `python bench.py "
               "--steps 200 --warmup 20 --no-cpu-baseline` (8 frames per step) on MI355X (profiles/collect.sh). Units KB -> bytes; "
               "FETCH doubled per MI355X_MICROARCH.md §HBM (gfx950 counts 128-B requests at 64 B); that correction is calibrated for wide "
               "coalesced streams, not for this kernel's 16-B gathers, and WRITE_SIZE counts the 4-B scattered hit records as partial "
               "lines, so treat the value as an estimate. Launch-weighted mean per k_trace launch."}
for wl, suffix in (("config2_100k_400x200_1pass", ""), ("config3_1M_400x200_4pass", "_c3"), ("target_10M_400x200_4pass", "_t")):
    f = os.path.join(here, "%s%s_pmc_summary.json" % (tag, suffix))
    if not os.path.exists(f):
        continue
    d = json.load(open(f))
    fetch = write = n = 0.0
    for k, v in d.items():
        if "k_trace" not in k or ", true, " in k:      # <FIRST, STATS, SPILL>: skip the stats build
            continue
        l = v["FETCH_SIZE"]["launches"]
        fetch += v["FETCH_SIZE"]["mean_per_launch"] * l
        write += v["WRITE_SIZE"]["mean_per_launch"] * l
        n += l
    out[wl] = {"source": "profiles/%s%s_pmc_summary.json" % (tag, suffix),
               "k_trace_FETCH_SIZE_KB": round(fetch / n, 1), "k_trace_WRITE_SIZE_KB": round(write / n, 1),
               "k_trace_hbm_bytes_per_launch": int(round((2 * fetch + write) / n * 1024))}
json.dump(out, open(os.path.join(here, "roofline_traffic.json"), "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if k != "_how"}, indent=1))
