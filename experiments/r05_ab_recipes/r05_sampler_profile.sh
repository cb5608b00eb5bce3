# Sampler kernel stats on the final tree (B = 1, N = 1 and N = 6; HIP-graph replay): bash experiments/r05_ab_recipes/r05_sampler_profile.sh
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/r05_samp; rm -rf "$OUT"; mkdir -p "$OUT"
for n in 1 6; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/s$n" -- python3 tools/prof_sampler.py 1 $n graph 250 > "$OUT/sampler_b1n$n.txt" 2> "$OUT/s$n.log"
  cp $(find "$OUT/s$n" -name "*kernel_stats.csv" | head -1) "$OUT/sampler_b1n${n}_kernel_stats.csv"; rm -rf "$OUT/s$n"
done
ls -la "$OUT"
