# 32-query training attention kernel against the 128-query / key-split ones (VF_ATTN_Q32=0): bash experiments/r05_ab_recipes/r05_attn_q32_ab.sh
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
{
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -k "attention" 2>&1 | tail -3
for s in 32 64 96; do python tools/attn_stamps.py $s q32 2>&1 | grep -v amdgpu.ids; done
for v in 0 1 0 1; do echo "== VF_ATTN_Q32=$v"; VF_ATTN_Q32=$v VF_ATTN_Q32_MIN=17 timeout 300 python tools/one_attn.py 2>&1 | grep "L=256" | grep -v -E "S=  (1|6) |S= 12 "; done
} > gpurun_out/r05_attn_q32.txt 2>&1
cat gpurun_out/r05_attn_q32.txt
