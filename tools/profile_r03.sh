#!/usr/bin/env bash
# The judged profile of round 3: rocprofv3 over the DEFAULT bench command (python3 bench.py: the mixed-precision
# headline with its f64 and f32 legs) — one plain run, one --kernel-trace --stats pass over the same command — and, per
# precision (mixed / f64 / f32, each as `--precision P --alt-steps 0 --no-cpu-baseline`: the timed steps are the same
# ones, the legs they skip are not profiled), FETCH_SIZE, WRITE_SIZE, TCC hit/miss, two SQ passes and GRBM in separate
# PMC passes (counters are never combined with tracing domains other than --kernel-trace).
set -u
root="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
out="$root/gpurun_out/prof_r03"
mkdir -p "$out"
export TMPDIR=/tmp
cd /tmp
# usage: tools/profile_r03.sh [default] [mixed] [f64] [f32] [summary]   (no arguments: everything; the whole set takes
# ~15 minutes of GPU time, so it can be split over several gpurun calls — gpurun_out/ is merged back between them)
what="${*:-default mixed f64 f32 summary}"
if [[ " $what " == *" default "* ]]; then
echo "--- plain run"; python3 "$root/bench.py" > "$out/bench_plain.log" 2>&1; tail -n 1 "$out/bench_plain.log" | cut -c1-300
echo "--- stats"; rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats" -- python3 "$root/bench.py" > "$out/stats.log" 2>&1; tail -n 1 "$out/stats.log" | cut -c1-120
fi
for prec in mixed f64 f32; do
  [[ " $what " == *" $prec "* ]] || continue
  pass() { local name="$1"; shift; rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$out/${prec}_$name" -- python3 "$root/bench.py" --precision $prec --alt-steps 0 --no-cpu-baseline > "$out/${prec}_$name.log" 2>&1; echo "pass $prec $name rc=$?"; }
  pass fetch FETCH_SIZE
  pass write WRITE_SIZE
  pass tcc TCC_HIT_sum TCC_MISS_sum
  pass sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
  pass sq2 SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM
  pass grbm GRBM_GUI_ACTIVE
done
if [[ " $what " == *" summary "* ]]; then
cd "$root" && python3 tools/profile_summary_r03.py "$out" > "$out/summary.txt" 2>&1
head -60 "$out/summary.txt"
fi
