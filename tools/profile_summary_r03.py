#!/usr/bin/env python3
"""Condense the rocprofv3 CSVs of tools/profile_r03.sh: kernel stats of the default command, and per precision the PMC
counters of the trace / shade kernels summed over dispatches.  Writes traffic.json = what bench.py's roofline objects
read (fabric bytes per launch, VALU-active share, L2 hit rate of the dominant closest-hit kernel of each precision)."""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

out = sys.argv[1]
N_SIMD = 1024  # 256 CUs x 4


def short(name):
    name = re.sub(r"\(.*", "", name)
    return name.replace("void ", "").replace("tk::", "")[:72]


def find(pattern):
    return sorted(glob.glob(os.path.join(out, pattern), recursive=True))


print(f"# rocprofv3 summary of {out} (tools/profile_r03.sh)")
stats = find("stats/**/*kernel_stats.csv")
kernel_avg_ns = {}
if stats:
    print("\n## kernel stats of the default command (--kernel-trace --stats; mixed-precision headline + f64 and f32 legs)")
    print(f"{'kernel':74s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>10s} {'pct':>6s}")
    for r in csv.DictReader(open(stats[0])):
        print(f"{short(r['Name']):74s} {r['Calls']:>7s} {float(r['TotalDurationNs'])/1e6:10.3f} "
              f"{float(r['AverageNs'])/1e3:10.1f} {float(r['Percentage']):6.2f}")
        kernel_avg_ns[r["Name"]] = float(r["AverageNs"])


def counters(prec, grp):
    files = find(f"{prec}_{grp}/**/*counter_collection.csv")
    acc = defaultdict(lambda: defaultdict(float))
    disp = defaultdict(set)
    dur = defaultdict(float)
    if not files:
        return acc, disp, dur
    for r in csv.DictReader(open(files[0])):
        k = short(r["Kernel_Name"])
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Dispatch_Id"] not in disp[k]:
            disp[k].add(r["Dispatch_Id"])
            if "Start_Timestamp" in r and "End_Timestamp" in r:
                dur[k] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    if not any(dur.values()):  # (no timestamps in the counter CSV: take them from the kernel trace of the same pass)
        for tf in find(f"{prec}_{grp}/**/*kernel_trace.csv"):
            for r in csv.DictReader(open(tf)):
                dur[short(r["Kernel_Name"])] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    return acc, disp, dur


traffic = {}
# dominant kernel per precision: the closest-hit instance that runs most of the rounds
# (round 0 of a render runs the CameraIo instance of the same kernel — the launch makes the camera rays itself; in a
# mixed render that launch belongs to the f64 rounds, so the f32 figure is the PathIo instance alone)
KEY = {"mixed": "k_trace_group<float, 1, false, false, PathIo<float>", "f64": "k_trace_group<double, 1, false, false, PathIo<double>",
       "f32": "k_trace_group<float, 1, false, false, PathIo<float>"}
ALSO = {"mixed": None, "f64": "k_trace_group<double, 1, false, false, CameraIo<double>", "f32": "k_trace_group<float, 1, false, false, CameraIo<float>"}
for prec in ("mixed", "f64", "f32"):
    print(f"\n# ===== precision {prec}: python3 bench.py --precision {prec} --alt-steps 0 --no-cpu-baseline")
    rec = {}
    for grp in ("fetch", "write", "tcc", "sq1", "sq2", "grbm"):
        acc, disp, dur = counters(prec, grp)
        if not acc:
            continue
        print(f"\n## counters: {grp} (sum over dispatches; ms = kernel time inside this pass)")
        for k in sorted(acc, key=lambda k: -sum(acc[k].values())):
            if not any(t in k for t in ("k_trace_group", "k_shade", "k_convert", "k_accumulate", "k_generate")):
                continue
            cs = "  ".join(f"{c}={v:.4g}" for c, v in sorted(acc[k].items()))
            print(f"{k:74s} n={len(disp[k]):5d} ms={dur[k]/1e6:9.2f}  {cs}")
            if k.startswith(KEY[prec]) or (ALSO[prec] and k.startswith(ALSO[prec])):
                for c, v in acc[k].items():
                    rec[c] = rec.get(c, 0.0) + v
                rec[f"{grp}_launches"] = rec.get(f"{grp}_launches", 0) + len(disp[k])
                rec[f"{grp}_ns"] = rec.get(f"{grp}_ns", 0.0) + dur[k]
    if "FETCH_SIZE" in rec and rec.get("fetch_launches"):
        # FETCH_SIZE / WRITE_SIZE are in KiB; 64-byte gathers are counted at 0.999 x bytes (profiles/r02_fetch_calibration.txt)
        t = {"kernel": KEY[prec] + ", true> (closest hit, one ray per lane, 64-byte compressed nodes)",
             "launches": rec["fetch_launches"],
             "fetch_bytes_per_launch": rec["FETCH_SIZE"] * 1024 / rec["fetch_launches"],
             "write_bytes_per_launch": rec.get("WRITE_SIZE", 0.0) * 1024 / max(rec.get("write_launches", 1), 1),
             "fetch_correction": 1.0,
             "fetch_correction_source": "profiles/r02_fetch_calibration.txt (64-byte random gathers: FETCH_SIZE = 0.999 x bytes)"}
        t["hbm_bytes_per_launch"] = t["fetch_bytes_per_launch"] + t["write_bytes_per_launch"]
        if "TCC_HIT_sum" in rec:
            t["l2_hit_rate"] = rec["TCC_HIT_sum"] / max(rec["TCC_HIT_sum"] + rec.get("TCC_MISS_sum", 0.0), 1.0)
        if "SQ_ACTIVE_INST_VALU" in rec and rec.get("sq2_ns"):
            clock = None
            if "GRBM_GUI_ACTIVE" in rec and rec.get("grbm_ns"):
                clock = rec["GRBM_GUI_ACTIVE"] / 8.0 / (rec["grbm_ns"] * 1e-9)  # sum over the 8 XCDs
                t["clock_GHz"] = clock / 1e9
            clock = clock or 2.3e9
            # SQ_ACTIVE_INST_VALU counts quad-cycles in which a VALU instruction of some wave is executing
            t["valu_busy"] = rec["SQ_ACTIVE_INST_VALU"] * 4.0 / (N_SIMD * rec["sq2_ns"] * 1e-9 * clock)
            t["valu_insts_per_launch"] = rec.get("SQ_INSTS_VALU", 0.0) / max(rec.get("sq2_launches", 1), 1)
            t["vmem_rd_insts_per_launch"] = rec.get("SQ_INSTS_VMEM_RD", 0.0) / max(rec.get("sq2_launches", 1), 1)
        if "SQ_WAVE_CYCLES" in rec:
            t["wave_time_split"] = {"waiting_on_memory_or_barrier": rec.get("SQ_WAIT_ANY", 0.0) / rec["SQ_WAVE_CYCLES"],
                                    "issue_stalled": rec.get("SQ_WAIT_INST_ANY", 0.0) / rec["SQ_WAVE_CYCLES"],
                                    "issuing": rec.get("SQ_ACTIVE_INST_ANY", 0.0) / rec["SQ_WAVE_CYCLES"]}
        traffic[prec] = t
with open(os.path.join(out, "traffic.json"), "w") as f:
    json.dump(traffic, f, indent=1)
print("\n## traffic.json\n" + json.dumps(traffic, indent=1))
