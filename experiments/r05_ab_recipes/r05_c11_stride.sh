# Is the short-K 1x1 conv on the 64x64 maps slow because of its channel stride (16 KB)?  The same bytes and flops at other map sizes.
cd "$(dirname "$0")/.."; mkdir -p gpurun_out
{
for cfg in "128 64 64 96" "128 64 32 384" "128 64 16 1536" "128 64 128 24" "64 128 64 96" "64 128 32 384" "64 128 16 1536" "192 64 64 96" "192 64 32 384"; do
  python tools/one_conv1x1.py $cfg 2>&1 | grep "TF"
done
} > gpurun_out/r05_c11_stride.txt 2>&1
cat gpurun_out/r05_c11_stride.txt
