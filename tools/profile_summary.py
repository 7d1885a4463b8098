#!/usr/bin/env python3
"""Condense the rocprofv3 CSVs of tools/profile.sh into one text summary (per kernel: calls, total/avg time,
and the PMC counters summed over dispatches)."""
import csv
import glob
import os
import re
import sys
from collections import defaultdict

out = sys.argv[1]


def short(name):
    name = re.sub(r"\(.*", "", name)
    name = name.replace("void ", "").replace("tk::", "")
    return name[:60]


def find(pattern):
    return sorted(glob.glob(os.path.join(out, pattern), recursive=True))


stats = find("stats/**/*kernel_stats.csv")
print(f"# rocprofv3 summary of {out}")
if stats:
    print("\n## kernel stats (--kernel-trace --stats)")
    rows = list(csv.DictReader(open(stats[0])))
    print(f"{'kernel':62s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>10s} {'pct':>6s}")
    for r in rows:
        print(f"{short(r['Name']):62s} {r['Calls']:>7s} {float(r['TotalDurationNs'])/1e6:10.3f} "
              f"{float(r['AverageNs'])/1e3:10.1f} {float(r['Percentage']):6.2f}")
for grp in ("fetch", "write", "tcc", "sq1", "sq2", "grbm"):
    files = find(f"{grp}/**/*counter_collection.csv")
    if not files:
        continue
    acc = defaultdict(lambda: defaultdict(float))
    ndisp = defaultdict(set)
    for r in csv.DictReader(open(files[0])):
        k = short(r["Kernel_Name"])
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        ndisp[k].add(r["Dispatch_Id"])
    print(f"\n## counters: {grp} (sum over dispatches)")
    for k in sorted(acc, key=lambda k: -sum(acc[k].values())):
        cs = "  ".join(f"{c}={v:.4g}" for c, v in sorted(acc[k].items()))
        print(f"{k:62s} n={len(ndisp[k]):5d}  {cs}")
