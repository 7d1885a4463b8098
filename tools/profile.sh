#!/usr/bin/env bash
# rocprofv3 passes over one bench.py run (on the GPU box).  usage: tools/profile.sh <tag> [bench args...]
# Kernel-trace/stats and each PMC group run as separate passes (counters are never combined with tracing domains
# other than --kernel-trace).  Raw output under gpurun_out/prof_<tag>/; tools/profile_summary.py condenses it.
set -u
tag="$1"; shift
root="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
out="$root/gpurun_out/prof_$tag"
mkdir -p "$out"
export TMPDIR=/tmp
cd /tmp
run() { # name, rocprof args...
  local name="$1"; shift
  echo "--- pass $name"
  rocprofv3 "$@" --output-format csv -d "$out/$name" -- python3 "$root/bench.py" "${BENCH_ARGS[@]}" > "$out/$name.log" 2>&1
  echo "rc=$? $(tail -n 1 "$out/$name.log" | cut -c1-200)"
}
BENCH_ARGS=("$@")
run stats --kernel-trace --stats
run fetch --kernel-trace --pmc FETCH_SIZE
run write --kernel-trace --pmc WRITE_SIZE
run tcc --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum
run sq1 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
run sq2 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM
run grbm --kernel-trace --pmc GRBM_GUI_ACTIVE
cd "$root" && python3 tools/profile_summary.py "$out" > "$out/summary.txt" 2>&1
cat "$out/summary.txt"
