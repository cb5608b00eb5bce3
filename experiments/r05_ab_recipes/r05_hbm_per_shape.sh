cd "$(dirname "$0")/../.." || exit 1; mkdir -p gpurun_out
{ for s in "64 64 64" "128 64 64" "64 128 32" "256 128 32" "192 192 16" "384 192 16"; do bash tools/hbm_per_shape.sh $s; done; } > gpurun_out/r05_hbm_per_shape.txt 2>&1
cat gpurun_out/r05_hbm_per_shape.txt
