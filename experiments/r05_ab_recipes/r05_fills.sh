cd "$(dirname "$0")/.."; export TMPDIR=/tmp; mkdir -p gpurun_out; rm -rf /tmp/ktrace
timeout -k 10 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/ktrace -- python3 bench.py --steps 3 --warmup 2 --no-sampler --no-cpu-baseline --no-roofline > /dev/null 2> gpurun_out/r05_fills.err
python3 tools/fills_in_step.py /tmp/ktrace > gpurun_out/r05_fills.txt 2>&1
cat gpurun_out/r05_fills.txt | head -60
