#!/bin/bash
# Evidence run on the final tree of a round: rocprofv3 kernel stats of the bench command, then FETCH_SIZE / WRITE_SIZE in two
# separate --pmc passes (never together with a trace domain other than --kernel-trace), summarised into the per-kernel
# roofline table.      bash tools/final_profile.sh <round, e.g. r06> <commit>
# Output: gpurun_out/<round>_final/{kernel_stats.csv, bench_profiled.json, pmc_traffic_kib_per_launch.json,
# kernel_roofline_table.md}; copy what is to be judged into profiles/ as <round>_bench_n1_train_kernel_stats.csv,
# <round>_bench_n1_profiled.json, <round>_bench_pmc_traffic_kib_per_launch.json, <round>_kernel_roofline_table.md.
set -u
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
R=${1:-rXX}
OUT=gpurun_out/${R}_final; rm -rf "$OUT"; mkdir -p "$OUT"
export VF_PMC_COMMIT=${2:-unknown}
export VF_PMC_COMMAND="python3 bench.py --steps 5 --warmup 2 --no-sampler --no-cpu-baseline"
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 bench.py --steps 20 --warmup 5 --no-sampler --no-cpu-baseline > "$OUT/bench_profiled.json" 2> "$OUT/stats.log"
cp "$(find "$OUT/stats" -name "*kernel_stats.csv" | head -1)" "$OUT/kernel_stats.csv"
rm -rf "$OUT/stats"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$OUT/pmc_$c" -- python3 bench.py --steps 5 --warmup 2 --no-sampler --no-cpu-baseline > /dev/null 2> "$OUT/pmc_$c.log" || echo "PMC pass $c failed (rc $?)" >> "$OUT/pmc_failed.txt"
done
python3 tools/pmc_summary.py "$OUT/pmc_traffic_kib_per_launch.json" "$OUT/pmc_FETCH_SIZE" "$OUT/pmc_WRITE_SIZE" > "$OUT/pmc_summary.txt" 2>&1
python3 tools/roofline_table.py "$OUT/kernel_stats.csv" "$OUT/pmc_traffic_kib_per_launch.json" 29 > "$OUT/kernel_roofline_table.md" 2>&1
rm -rf "$OUT/pmc_FETCH_SIZE" "$OUT/pmc_WRITE_SIZE"
ls -la "$OUT"
