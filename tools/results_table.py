#!/usr/bin/env python3
"""Markdown result table from the bench lines profiles/<tag>_bench_<workload>.json.  usage: tools/results_table.py r04"""
import json, os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
W = [("2. 400x200, 1 pass, 100k tris", "config2_100k_400x200_1pass"), ("3. 400x200, 4 passes, 1M tris", "config3_1M_400x200_4pass"),
     ("**north-star target**: 400x200, 4 passes, 10M tris", "target_10M_400x200_4pass"),
     ("4. 400x1000, 4 passes, 10M tris", "config4_10M_400x1000_4pass"), ("5. 400x1000, 8 passes, per-triangle materials, Cook-Torrance lobe", "config5_10M_400x1000_8pass_pertri")]
print("| config | CPU img/s (threads of usable CPUs) | 1x MI355X img/s (`value`: through the last D2H copy) | images left in HBM (`hbm_resident`) | 1 pose per launch set | one synchronous `rr_simulate` | wave-passes/s | dominant kernel: VALU issue frac alone / live / useful share | launch alone / live | strong-scaling bound of ONE frame at 2 / 4 / 8 GPUs |")
print("|---|---|---|---|---|---|---|---|---|---|")
for label, w in W:
    f = os.path.join(R, "profiles", "%s_bench_%s.json" % (tag, w))
    if not os.path.exists(f):
        continue
    d = json.load(open(f)); r = d["roofline"]; cb = d.get("cpu_baseline") or {}
    hr, sp, sf, px = (d.get("hbm_resident") or d.get("host_resident") or {}), d.get("single_pose") or {}, d.get("single_frame_sync") or {}, d.get("strong_scaling_proxy") or {}
    fr = lambda x: "-" if x is None else "%.2f" % x
    u = (r.get("useful_issue_frac") or {}).get("value")
    print("| %s | %s (%s of %s) | **%s** | %s | %s | %s | %.2f G | `%s` %s / %s / %s | %.0f / %.0f us | %s |" % (
        label, cb.get("value", "-"), cb.get("cores", "-"), cb.get("usable_cpus", "-"), "{:,.0f}".format(d["value"]),
        "{:,.0f}".format(hr["value"]) if hr else "-", "{:,.0f}".format(sp["value"]) if sp else "-",
        ("%.3f ms" % sf["ms_per_frame"]) if sf else "-", d["rays_per_s"] / 1e9, r["kernel"].split(" ")[0], fr(r["isolated"]["frac"]), fr(r["frac"]), fr(u),
        r["isolated"]["avg_launch_us"], r["avg_launch_us"],
        " / ".join("%.2fx" % px[k] for k in ("speedup_bound_2_gpus", "speedup_bound_4_gpus", "speedup_bound_8_gpus")) if px else "-"))
