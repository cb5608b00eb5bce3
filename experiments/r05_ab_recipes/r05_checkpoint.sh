# full GPU suite + the driver's bench command (what the round-end run does) + the other configurations, on one box
cd "$(dirname "$0")/../.." || exit 1
python -m pytest tests -m gpu -q -s --durations=15 -p no:cacheprovider > gpurun_out/r05_gputest_final.log 2>&1; echo rc=$? >> gpurun_out/r05_gputest_final.log
tail -3 gpurun_out/r05_gputest_final.log
python bench.py > gpurun_out/r05_bench_final.json 2> gpurun_out/r05_bench_final.err; tail -c 300 gpurun_out/r05_bench_final.json
python bench.py --batch 8 --no-cpu-baseline --no-sampler --steps 20 --warmup 5 > gpurun_out/r05_bench_c4_b8n6.json 2>/dev/null
python bench.py --ragged --no-cpu-baseline --no-sampler --steps 20 --warmup 5 > gpurun_out/r05_bench_ragged_b16n6.json 2>/dev/null
python bench.py --batch 32 --no-cpu-baseline --no-sampler --steps 10 --warmup 5 > gpurun_out/r05_bench_b32n6.json 2>/dev/null
for f in c4_b8n6 ragged_b16n6 b32n6; do python -c "
import json,sys
d=json.loads(open('gpurun_out/r05_bench_$f.json').read().strip().splitlines()[-1]); print('$f', round(d['ms_per_step'],2), 'ms', round(d['value'],1), '/s')"; done
