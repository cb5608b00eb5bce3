# Round-5 evidence run on the GPU box (after the code is final):  bash experiments/r05_ab_recipes/r05_final_profile.sh <commit>
#   1. rocprofv3 --kernel-trace --stats of the bench command            -> gpurun_out/r05_final/kernel_stats.csv
#   2. FETCH_SIZE / WRITE_SIZE in two separate --pmc passes (--kernel-trace only beside them) -> pmc json
#   3. SQ counters of the four MFMA-bound shapes of profiles/r04_sq_counters.txt
#   4. the sampler's kernel stats at B = 1 N = 1
# The program after `--` is python3 itself (no env / bash -c hop: rocprofv3's library has initialised the GPU).
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/r05_final; rm -rf "$OUT"; mkdir -p "$OUT"
export VF_PMC_COMMIT=${1:-unknown}
export VF_PMC_COMMAND="python3 bench.py --steps 5 --warmup 2 --no-sampler --no-cpu-baseline"
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 bench.py --steps 20 --warmup 5 --no-sampler --no-cpu-baseline > "$OUT/bench_profiled.json" 2> "$OUT/stats.log"
cp $(find "$OUT/stats" -name "*kernel_stats.csv" | head -1) "$OUT/kernel_stats.csv"
rm -rf "$OUT/stats"
{
echo "== F(4x4) forward kernel, 64 -> 64 @ 64x64, S = 96";   bash tools/sq_collect.sh "$OUT/sq" wino44_conv -- python3 tools/one_wino.py 64 64 64
echo "== F(4x4) forward kernel, 192 -> 64 @ 64x64";          bash tools/sq_collect.sh "$OUT/sq" wino44_conv -- python3 tools/one_wino.py 192 64 64
echo "== nested kernel, 192 -> 192 @ 16x16";                 bash tools/sq_collect.sh "$OUT/sq" wino_conv_kernel -- python3 tools/one_wino.py 192 192 16
echo "== F(4x4) weight-gradient kernel, 192 -> 64 @ 64x64";  bash tools/sq_collect.sh "$OUT/sq" wino44_wgrad -- python3 tools/one_wgrad.py 192 64 64
echo "== F(4x4) weight-gradient kernel, 320 -> 320 @ 8x8";   bash tools/sq_collect.sh "$OUT/sq" wino44_wgrad -- python3 tools/one_wgrad.py 320 320 8
} > "$OUT/sq_counters.txt" 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/samp" -- python3 tools/prof_sampler.py 1 1 graph 250 > "$OUT/sampler_b1n1.txt" 2> "$OUT/samp.log"
cp $(find "$OUT/samp" -name "*kernel_stats.csv" | head -1) "$OUT/sampler_b1n1_kernel_stats.csv"; rm -rf "$OUT/samp"
for c in FETCH_SIZE WRITE_SIZE; do
  # (a counter pass serialises the kernels: the short form of the command, as in rounds 1-4; bounded -- a pass that dies
  #  inside the profiler must not take the rest of the call with it)
  timeout -k 10 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$OUT/pmc_$c" -- python3 bench.py --steps 5 --warmup 2 --no-sampler --no-cpu-baseline > /dev/null 2> "$OUT/pmc_$c.log" || echo "PMC pass $c failed (rc $?)" >> "$OUT/pmc_failed.txt"
done
python3 tools/pmc_summary.py "$OUT/pmc_traffic_kib_per_launch.json" "$OUT/pmc_FETCH_SIZE" "$OUT/pmc_WRITE_SIZE" > "$OUT/pmc_summary.txt" 2>&1
python3 tools/roofline_table.py "$OUT/kernel_stats.csv" "$OUT/pmc_traffic_kib_per_launch.json" 29 > "$OUT/kernel_roofline_table.md" 2>&1
rm -rf "$OUT/pmc_FETCH_SIZE" "$OUT/pmc_WRITE_SIZE"
ls -la "$OUT"
