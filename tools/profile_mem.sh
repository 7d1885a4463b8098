#!/usr/bin/env bash
# memory-path PMC passes (TA / TCP) over a short bench run.  usage: tools/profile_mem.sh <tag> [bench args]
set -u
tag="$1"; shift
root="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
out="$root/gpurun_out/prof_$tag"; mkdir -p "$out"
export TMPDIR=/tmp; cd /tmp
BENCH_ARGS=("$@")
run() { local name="$1"; shift; echo "start $name"; timeout -k 5 120 rocprofv3 "$@" --output-format csv -d "$out/$name" -- python3 "$root/bench.py" "${BENCH_ARGS[@]}" > "$out/$name.log" 2>&1; echo "pass $name rc=$?"; }
run tcp1 --kernel-trace --pmc TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TOTAL_CACHE_ACCESSES_sum
run tcp2 --kernel-trace --pmc TCP_TCP_LATENCY_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TOTAL_READ_sum
# (a TA_* pass was tried too: rocprofv3 never finished it on this pool — killed by the 120 s timeout — so it is not run)
cd "$root" && python3 - "$out" <<'PY'
import csv, glob, os, sys
from collections import defaultdict
out = sys.argv[1]
for grp in ("ta", "tcp1", "tcp2"):
    fs = glob.glob(os.path.join(out, grp, "**", "*counter_collection.csv"), recursive=True)
    if not fs: continue
    acc = defaultdict(lambda: defaultdict(float))
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"].replace("void tk::", "")[:48]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    print("##", grp)
    for k in sorted(acc, key=lambda k: -sum(acc[k].values()))[:4]:
        print(f"{k:50s}", "  ".join(f"{c}={v:.4g}" for c, v in sorted(acc[k].items())))
PY
