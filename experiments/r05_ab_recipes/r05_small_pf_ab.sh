cd "$(dirname "$0")/../.." || exit 1
python -m pytest tests/test_gpu_kernels.py -q -x -k "conv_small or folded_residual or groupnorm_without" -p no:cacheprovider 2>&1 | tail -2
for pf in 0 1; do echo "== VF_SMALL_PF=$pf"; VF_SMALL_PF=$pf python tools/small_conv_rounds.py 2>/dev/null; done
