"""Summarise a rocprofv3 kernel_trace.csv: per-kernel durations and stream overlap."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows = [r for r in rows if "rr::" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = len(rows)
rows = rows[n // 2: n // 2 + 4000]           # steady-state slice
t0 = int(rows[0]["Start_Timestamp"]); t1 = max(int(r["End_Timestamp"]) for r in rows)
dur = collections.defaultdict(list)
ev = []
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    dur[r["Kernel_Name"].split("(")[0][-28:]].append(e - s)
    ev.append((s, 1)); ev.append((e, -1))
ev.sort()
busy = collections.Counter(); cur = 0; last = t0
for t, d in ev:
    busy[cur] += t - last; last = t; cur += d
tot = t1 - t0
print("window %.2f ms, kernels %d" % (tot / 1e6, len(rows)))
for k, v in dur.items():
    print("  %-30s n=%5d avg %.1f us  sum %.1f%% of window" % (k, len(v), sum(v) / len(v) / 1e3, 100.0 * sum(v) / tot))
print("  concurrency histogram (share of window):", {k: round(100.0 * v / tot, 1) for k, v in sorted(busy.items())})
