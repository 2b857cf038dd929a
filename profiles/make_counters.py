#!/usr/bin/env python3
"""profiles/roofline_counters.json from the PMC summaries profiles/collect.sh writes -- what bench.py divides by
the launch durations it measures live:
  SQ_INSTS_VALU   wave-level VALU instructions per launch of each kernel (the VALU-issue roofline)
  hbm_bytes       2 x FETCH_SIZE + WRITE_SIZE (KB -> B) per launch: HBM traffic (secondary view)
The k_trace variants are <FIRST, STATS, SPILL, CULL, STACKLESS> (the fifth since round 6; false unless RR_STACKLESS=1): "trace0" = <true, false, *, *> (pass 0), "trace" = <false, false, *, *>
(passes 1..P-1, launch-weighted mean); the counting builds (<*, true, *, *>) are instrumentation and left out.
usage: profiles/make_counters.py <round tag> [workload=suffix ...]   e.g.  make_counters.py r03 target_10M_400x200_4pass= config3_1M_400x200_4pass=_c3"""
import json, os, sys
here = os.path.dirname(os.path.abspath(__file__))
tag = sys.argv[1]
pairs = [a.split("=") for a in sys.argv[2:]] or [["target_10M_400x200_4pass", ""], ["config3_1M_400x200_4pass", "_c3"], ["config2_100k_400x200_1pass", "_c2"]]
path = os.path.join(here, "roofline_counters.json")
out = json.load(open(path)) if os.path.exists(path) else {}
out["_how"] = ("rocprofv3 --pmc <counters> --kernel-trace (separate passes: FETCH_SIZE, WRITE_SIZE, SQ_*, TCC/TCP, LDS) of `python bench.py "
               "--steps 60 --warmup 5 --no-cpu-baseline --no-extras [--workload W]` on MI355X (profiles/collect.sh); means per launch. "
               "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 tallies 128-B requests at 64 B); that correction is calibrated for wide "
               "coalesced streams, not for 16-B gathers, and WRITE_SIZE counts 4-B scattered stores as partial lines: hbm_bytes is an estimate.")


sys.path.insert(0, os.path.dirname(here))
from bench import kernel_source_hash        # the sources the counters were collected from (run this right after collect.sh)
src_hash = kernel_source_hash()


def classify(name):
    n = name.replace("void ", "").replace("rr::", "")
    if n.startswith("k_trace_repair"):
        return None        # (round 5) the remainder launch behind a tightened trace launch: empty unless a segment overflowed its row
    if n.startswith("k_trace"):
        args = n[n.index("<") + 1:n.index(">")].replace(" ", "").split(",")
        if args[1] == "true":
            return None
        return "trace0" if args[0] == "true" else "trace"
    for k in ("shade", "scan", "column", "assemble"):
        if n.startswith("k_" + k):
            return k
    return None


for wl, suffix in pairs:
    f = os.path.join(here, "%s%s_pmc_summary.json" % (tag, suffix))
    if not os.path.exists(f):
        continue
    d = json.load(open(f))
    acc = {}
    for kname, v in d.items():
        k = classify(kname)
        if k is None:
            continue
        a = acc.setdefault(k, {})
        for c, x in v.items():
            t = a.setdefault(c, [0.0, 0])
            t[0] += x["mean_per_launch"] * x["launches"]; t[1] += x["launches"]
    kernels = {}
    for k, a in acc.items():
        m = {c: t[0] / t[1] for c, t in a.items() if t[1]}
        e = {"SQ_INSTS_VALU": int(round(m.get("SQ_INSTS_VALU", 0))), "launches_profiled": int(max(t[1] for t in a.values()))}
        if "FETCH_SIZE" in m and "WRITE_SIZE" in m:
            e["FETCH_SIZE_KB"] = round(m["FETCH_SIZE"], 1); e["WRITE_SIZE_KB"] = round(m["WRITE_SIZE"], 1)
            e["hbm_bytes"] = int(round((2 * m["FETCH_SIZE"] + m["WRITE_SIZE"]) * 1024))
        for c in ("SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_INSTS_VMEM_RD",
                  "SQ_INSTS_LDS", "SQ_LDS_BANK_CONFLICT", "TCC_HIT_sum", "TCC_MISS_sum", "TCP_TOTAL_CACHE_ACCESSES_sum", "TCP_TCC_READ_REQ_sum"):
            if c in m:
                e[c] = int(round(m[c]))
        kernels[k] = e
    fpl = 8
    bf = os.path.join(here, "%s%s_bench_under_rocprof.json" % (tag, suffix))
    if os.path.exists(bf):
        try:
            b = json.load(open(bf)); fpl = int(b["config"]["frames_per_batch"]) // max(int(b["n_gpus"]), 1)
        except Exception:
            pass
    out[wl] = {"source": "profiles/%s%s_pmc_summary.json" % (tag, suffix), "frames_per_launch": fpl, "kernels": kernels,
               "kernel_source_sha16": src_hash}
json.dump(out, open(path, "w"), indent=1, sort_keys=True)
print(json.dumps({k: {kk: vv.get("SQ_INSTS_VALU") for kk, vv in v["kernels"].items()} for k, v in out.items() if k != "_how"}, indent=1))
