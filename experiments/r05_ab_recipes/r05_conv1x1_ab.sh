cd "$(dirname "$0")/../.." || exit 1
python -m pytest tests/test_gpu_kernels.py -q -x -k "conv1x1" -p no:cacheprovider 2>&1 | tail -3
echo "== round-4 policy (64-channel variant off)"; VF_CONV1X1_64=0 python tools/conv1x1_table.py 2>/dev/null | sed 's/ | bf16x3.*//' 
echo "== 64-channel variant forced wherever legal"; VF_CONV1X1_FORCE=1 VF_CONV1X1_NCW=1 python tools/conv1x1_table.py 2>/dev/null | sed 's/ | bf16x3.*//'
echo "== 128-channel variant forced wherever legal"; VF_CONV1X1_FORCE=1 VF_CONV1X1_NCW=2 python tools/conv1x1_table.py 2>/dev/null | sed 's/ | bf16x3.*//'
echo "== new default policy"; python tools/conv1x1_table.py 2>/dev/null | sed 's/ | bf16x3.*//'
