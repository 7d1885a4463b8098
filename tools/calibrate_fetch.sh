#!/usr/bin/env bash
# FETCH_SIZE calibration for the trace kernel's access pattern (on the GPU box): tools/ubench_gather calib under
# rocprofv3 --pmc FETCH_SIZE; prints counter / known bytes per kernel.  Output: gpurun_out/calib_fetch/summary.txt
set -u
root="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
out="$root/gpurun_out/calib_fetch"
mkdir -p "$out"
export TMPDIR=/tmp
cd /tmp
"$root/tools/ubench_gather" calib > "$out/plain.log" 2>&1 || { cat "$out/plain.log"; exit 1; }
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$out/fetch" -- "$root/tools/ubench_gather" calib > "$out/fetch.log" 2>&1
echo "rocprofv3 rc=$?"
python3 - "$out" <<'PY' | tee "$out/summary.txt"
import csv, glob, os, sys
out = sys.argv[1]
known = {}
for line in open(os.path.join(out, "plain.log")):
    if line.startswith("CALIB"):
        f = line.split()
        known[f[1]] = {"fetches": float(f[3]), "record": int(f[5]), "requested": float(f[7])}
files = sorted(glob.glob(os.path.join(out, "fetch", "**", "*counter_collection.csv"), recursive=True))
rows = [r for r in csv.DictReader(open(files[0])) if r["Counter_Name"] == "FETCH_SIZE" and "k_gather" in r["Kernel_Name"]]
print("# FETCH_SIZE calibration: dependent random gathers from a 1 GiB table (tools/ubench_gather calib)")
for r in rows:
    kib = float(r["Counter_Value"])
    name = "pair_128B" if "128" in r["Kernel_Name"].split("k_gather")[1][:12] else "pair_64B"
    k = known[name]
    b = kib * 1024
    print(f"{name}: FETCH_SIZE {b:.4g} B for {k['fetches']:.4g} fetches of {k['record']} B: {b / k['fetches']:.1f} B per fetch, "
          f"counter / requested bytes = {b / k['requested']:.3f}")
PY
