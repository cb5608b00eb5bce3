# 32-query attention kernel in the product: kernel + model tests, then the training step A/B (VF_ATTN_Q32=0/1)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
{
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -x -k "attention" 2>&1 | tail -3
timeout 1200 python -m pytest tests/test_gpu_model.py tests/test_gpu_step_graph.py -q -x 2>&1 | tail -3
bash tools/ab_env.sh VF_ATTN_Q32 0 1 2
} > gpurun_out/r05_attn_q32_step.txt 2>&1
cat gpurun_out/r05_attn_q32_step.txt
