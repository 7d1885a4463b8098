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

# ---- HBM-side traffic of the dominant kernel (closest-hit trace), per launch, for bench.py's roofline.traffic.
# FETCH_SIZE / WRITE_SIZE are in KiB (rocprofv3 derived counters: TCC_EA0_RDREQ*64 B etc.).  MI355X_MICROARCH.md:
# on gfx950 FETCH_SIZE under-reports a wide coalesced stream (128-byte requests tallied at 64 B) by exactly 2x and
# calls other patterns uncalibrated.  Calibrated for THIS kernel's pattern (profiles/r02_fetch_calibration.txt,
# tools/calibrate_fetch.sh: dependent random gathers from a 1 GiB table): a pair reading a 64-byte record — the
# compressed node, the primitive record, the first half of a path record, i.e. everything this kernel fetches —
# is counted at 63.9 B per fetch: factor 1.00, the raw counter is the byte count.  (128-byte records: 64.2 B per
# fetch, factor 0.50 — the guide's rule.)
FETCH_FACTOR_64B_GATHER = 1.0
import json


def counter_total(grp, counter, key):
    files = find(f"{grp}/**/*counter_collection.csv")
    if not files:
        return None, 0
    tot, disp = 0.0, set()
    for r in csv.DictReader(open(files[0])):
        if key in r["Kernel_Name"] and r["Counter_Name"] == counter:
            tot += float(r["Counter_Value"])
            disp.add(r["Dispatch_Id"])
    return tot, len(disp)


key = "false, false, tk::PathIo<float>"  # k_trace_group<float, G, ANY_HIT=false, COUNT=false, PathIo<float>>
fetch, nf = counter_total("fetch", "FETCH_SIZE", key)
write, nw = counter_total("write", "WRITE_SIZE", key)
if fetch is not None and nf:
    t = {"kernel": "tk::k_trace_group<float,1,false,false,PathIo<float>,true> (closest hit, one ray per lane, compressed nodes)", "launches": nf,
         "fetch_bytes_per_launch": fetch * 1024 / nf, "write_bytes_per_launch": (write or 0) * 1024 / max(nw, 1),
         "fetch_correction": FETCH_FACTOR_64B_GATHER,
         "fetch_correction_source": "profiles/r02_fetch_calibration.txt (64-byte random gathers: FETCH_SIZE = 0.999 x bytes)"}
    t["hbm_bytes_per_launch"] = t["fetch_bytes_per_launch"] + t["write_bytes_per_launch"]
    t["hbm_bytes_per_launch_corrected"] = t["fetch_bytes_per_launch"] / FETCH_FACTOR_64B_GATHER + t["write_bytes_per_launch"]
    with open(os.path.join(out, "traffic.json"), "w") as f:
        json.dump(t, f, indent=1)
    print("\n## traffic.json\n" + json.dumps(t, indent=1))
