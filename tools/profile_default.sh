#!/usr/bin/env bash
# The judged profile: rocprofv3 over the DEFAULT bench command (python3 bench.py): one plain run, one
# --kernel-trace --stats pass over the same command, and FETCH_SIZE / WRITE_SIZE in separate PMC passes (those with
# --f64-steps 0 --no-cpu-baseline: the f32 steps they measure are the same, the legs they skip are not profiled).
set -u
root="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
out="$root/gpurun_out/prof_default"
mkdir -p "$out"
export TMPDIR=/tmp
cd /tmp
echo "--- plain run"; python3 "$root/bench.py" > "$out/bench_plain.log" 2>&1; tail -n 1 "$out/bench_plain.log" | cut -c1-400
echo "--- stats"; rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats" -- python3 "$root/bench.py" > "$out/stats.log" 2>&1; tail -n 1 "$out/stats.log" | cut -c1-200
echo "--- fetch"; rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$out/fetch" -- python3 "$root/bench.py" --f64-steps 0 --no-cpu-baseline > "$out/fetch.log" 2>&1; tail -n 1 "$out/fetch.log" | cut -c1-120
echo "--- write"; rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$out/write" -- python3 "$root/bench.py" --f64-steps 0 --no-cpu-baseline > "$out/write.log" 2>&1; tail -n 1 "$out/write.log" | cut -c1-120
cd "$root" && python3 tools/profile_summary.py "$out" > "$out/summary.txt" 2>&1
# the raw per-dispatch CSVs are large: keep only the stats CSV and the summaries
find "$out" -name "*counter_collection.csv" -delete; find "$out" -name "*kernel_trace.csv" -delete
head -40 "$out/summary.txt"; tail -n 16 "$out/summary.txt"
