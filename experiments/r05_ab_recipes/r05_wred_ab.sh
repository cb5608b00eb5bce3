# Deferred slab sums of the Winograd weight gradients (one multi launch per backward pass): tests, then the step A/B
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
{
timeout 1500 python -m pytest tests/test_gpu_model.py tests/test_gpu_step_graph.py tests/test_gpu_two_rank.py -q 2>&1 | grep -E "passed|failed|error|Error" | tail -8
bash tools/ab_env.sh VF_WRED_DEFER 0 1 3
} > gpurun_out/r05_wred.txt 2>&1
cat gpurun_out/r05_wred.txt
