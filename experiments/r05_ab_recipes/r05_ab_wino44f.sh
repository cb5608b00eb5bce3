# A/B of two builds of the C ABI on one box.  build/ab/libvf_base*.so = the library of the commit BEFORE the change under test:
#   git stash; python -m view_fusion_amd.build; mkdir -p build/ab; cp view_fusion_amd/lib/libvf_hip.so build/ab/libvf_baseN.so; git stash pop; python -m view_fusion_amd.build
# (scratch files, removed after the round's measurements; results: profiles/r05_*.md)
set -x
cd "$(dirname "$0")/../.." || exit 1
python -m pytest tests/test_gpu_kernels.py -q -x -k "winograd44 or wino44" -p no:cacheprovider 2>&1 | tail -5
python -m pytest tests/test_gpu_model.py -q -x -k "extrapolate_real or unet_small_forward" -p no:cacheprovider 2>&1 | tail -3
echo "== base"; VF_DEBUG_AB=1 VF_HIP_LIB=$PWD/build/ab/libvf_base.so python tools/wino44_table.py 2>&1 | grep -v "^\[view" > gpurun_out/r05_w44_base.txt; tail -1 gpurun_out/r05_w44_base.txt
echo "== new"; python tools/wino44_table.py > gpurun_out/r05_w44_new.txt 2>&1; tail -1 gpurun_out/r05_w44_new.txt
echo "== base"; VF_DEBUG_AB=1 VF_HIP_LIB=$PWD/build/ab/libvf_base.so python tools/wino44_table.py 2>&1 | grep -v "^\[view" > gpurun_out/r05_w44_base2.txt; tail -1 gpurun_out/r05_w44_base2.txt
echo "== new"; python tools/wino44_table.py > gpurun_out/r05_w44_new2.txt 2>&1; tail -1 gpurun_out/r05_w44_new2.txt
bash tools/sq_collect.sh gpurun_out/sq1 wino44_conv -- python3 tools/one_wino.py 64 64 64 > gpurun_out/r05_sq_w44f_64_64_64.txt 2>&1
bash tools/sq_collect.sh gpurun_out/sq2 wino44_conv -- python3 tools/one_wino.py 192 64 64 > gpurun_out/r05_sq_w44f_192_64_64.txt 2>&1
cat gpurun_out/r05_sq_w44f_64_64_64.txt gpurun_out/r05_sq_w44f_192_64_64.txt
