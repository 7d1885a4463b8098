#!/usr/bin/env bash
# short counter profile: kernel stats + the two SQ passes.  usage: tools/profile_sq.sh <tag> [bench args...]
set -u
tag="$1"; shift
root="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
out="$root/gpurun_out/prof_$tag"
mkdir -p "$out"
export TMPDIR=/tmp
cd /tmp
run() { local name="$1"; shift; rocprofv3 "$@" --output-format csv -d "$out/$name" -- python3 "$root/bench.py" "${BENCH_ARGS[@]}" > "$out/$name.log" 2>&1; echo "pass $name rc=$?"; }
BENCH_ARGS=("$@")
run stats --kernel-trace --stats
run sq1 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
run sq2 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM
cd "$root" && python3 tools/profile_summary.py "$out" > "$out/summary.txt" 2>&1
find "$out" -name "*counter_collection.csv" -delete; find "$out" -name "*kernel_trace.csv" -delete
grep -E "k_trace_group|k_shade" "$out/summary.txt" | cut -c1-400
