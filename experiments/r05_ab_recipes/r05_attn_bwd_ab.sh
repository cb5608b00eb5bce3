# Attention backward: the two round-5 launches (VF_ATTN_DSCORE: dS + dQ in the 32-query kernel; VF_ATTN_DVDK: dV + dK in one
# dedicated kernel) against the four batched products + softmax backward of rounds 1-4
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
{
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -x -k "attention or linear or gemm" 2>&1 | grep -E "passed|failed|error" | tail -3
for cfg in "0 0" "1 0" "1 1" "0 0" "1 0" "1 1"; do set -- $cfg; echo "== VF_ATTN_DSCORE=$1 VF_ATTN_DVDK=$2"; VF_ATTN_DSCORE=$1 VF_ATTN_DVDK=$2 timeout 300 python tools/one_attn_bwd.py 48 96 2>&1 | grep "L=256"; done
timeout 1500 python -m pytest tests/test_gpu_model.py tests/test_gpu_step_graph.py -q -x 2>&1 | grep -E "passed|failed|error" | tail -3
for i in 1 2; do for cfg in "0 0" "1 1"; do set -- $cfg; VF_ATTN_DSCORE=$1 VF_ATTN_DVDK=$2 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-sampler --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('DSCORE=$1 DVDK=$2', round(d['ms_per_step'],3), 'ms/step')"; done; done
} > gpurun_out/r05_attn_bwd.txt 2>&1
cat gpurun_out/r05_attn_bwd.txt
