# In-kernel phase clocks of the 32-query attention kernel (1, 2, 3 workgroups per compute unit) and of the 128-query one
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
{
for s in 24 32 64 96; do python tools/attn_stamps.py $s q32; done
python tools/attn_stamps.py 96
} > gpurun_out/r05_attn_q32_stamps.txt 2>&1
cat gpurun_out/r05_attn_q32_stamps.txt
