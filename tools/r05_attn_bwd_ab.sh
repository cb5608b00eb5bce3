# Attention backward: fused dS kernel (VF_ATTN_DSCORE) and per-XCD batch placement of the batched products (VF_GEMM_XCD)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
{
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -x -k "attention or linear or gemm" 2>&1 | grep -E "passed|failed|error" | tail -3
for cfg in "0 0" "0 1" "1 0" "1 1" "0 0" "1 1"; do set -- $cfg; echo "== VF_ATTN_DSCORE=$1 VF_GEMM_XCD=$2"; VF_ATTN_DSCORE=$1 VF_GEMM_XCD=$2 timeout 300 python tools/one_attn_bwd.py 48 96 2>&1 | grep "L="; done
timeout 1500 python -m pytest tests/test_gpu_model.py tests/test_gpu_step_graph.py -q -x 2>&1 | grep -E "passed|failed|error" | tail -3
for i in 1 2; do for cfg in "0 0" "1 1"; do set -- $cfg; VF_ATTN_DSCORE=$1 VF_GEMM_XCD=$2 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-sampler --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('DSCORE=$1 XCD=$2', round(d['ms_per_step'],3), 'ms/step')"; done; done
} > gpurun_out/r05_attn_bwd.txt 2>&1
cat gpurun_out/r05_attn_bwd.txt
