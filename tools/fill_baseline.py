#!/usr/bin/env python3
"""Fills the R2* placeholders of BASELINE.md §4 from profiles/<tag>_bench_*.json.  usage: tools/fill_baseline.py r02c"""
import json, os, re, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
W = {"C2": "config2_100k_400x200_1pass", "C3": "config3_1M_400x200_4pass", "T": "target_10M_400x200_4pass",
     "C4": "config4_10M_400x1000_4pass", "C5": "config5_10M_400x1000_8pass_pertri"}
def load(w, suffix=""):
    f = os.path.join(R, "profiles", "%s_bench_%s%s.json" % (tag, w, suffix))
    return json.load(open(f)) if os.path.exists(f) else None
s = open(os.path.join(R, "BASELINE.md")).read()
for key, w in W.items():
    d, d1 = load(w), load(w, "_fpr1")
    if not d:
        continue
    r = d["roofline"]
    rep = {"R2%sF1" % key: ("{:,.0f}".format(d1["value"]) if d1 else ""),
           "R2%sFR" % key: "%.2f / %.2f" % (r["frac"], r["isolated"]["frac"]),
           "R2%sR" % key: "%.2f G" % (d["rays_per_s"] / 1e9),
           "R2%sL" % key: "%.0f us / %.0f us" % (r["avg_launch_us"], r["isolated"]["avg_launch_us"]),
           "R2%s" % key: "{:,.0f}".format(d["value"])}
    for k in sorted(rep, key=len, reverse=True):
        s = re.sub(r"\b%s\b" % k, rep[k], s)
open(os.path.join(R, "BASELINE.md"), "w").write(s)
print("filled from", tag)
