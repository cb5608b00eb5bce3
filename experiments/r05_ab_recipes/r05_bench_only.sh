cd "$(dirname "$0")/../.." || exit 1; mkdir -p gpurun_out
python bench.py > gpurun_out/r05_bench_final.json 2> gpurun_out/r05_bench_final.err; tail -c 200 gpurun_out/r05_bench_final.json
