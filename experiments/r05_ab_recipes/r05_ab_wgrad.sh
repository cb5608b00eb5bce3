# A/B of two builds of the C ABI on one box.  build/ab/libvf_base*.so = the library of the commit BEFORE the change under test:
#   git stash; python -m view_fusion_amd.build; mkdir -p build/ab; cp view_fusion_amd/lib/libvf_hip.so build/ab/libvf_baseN.so; git stash pop; python -m view_fusion_amd.build
# (scratch files, removed after the round's measurements; results: profiles/r05_*.md)
set -x
cd "$(dirname "$0")/../.." || exit 1
python -m pytest tests/test_gpu_kernels.py -q -x -k "wgrad or winograd" -p no:cacheprovider 2>&1 | tail -5
for i in 1 2; do
echo "== base"; VF_DEBUG_AB=1 VF_HIP_LIB=$PWD/build/ab/libvf_base.so python tools/wgrad_table.py 2>&1 | grep -v "^\[view" > gpurun_out/r05_wg_base$i.txt; tail -2 gpurun_out/r05_wg_base$i.txt
echo "== new"; python tools/wgrad_table.py > gpurun_out/r05_wg_new$i.txt 2>&1; tail -2 gpurun_out/r05_wg_new$i.txt
done
bash tools/sq_collect.sh gpurun_out/sq1 wino44_wgrad -- python3 tools/one_wgrad.py 192 64 64 > gpurun_out/r05_sq_wg_192_64_64.txt 2>&1
bash tools/sq_collect.sh gpurun_out/sq2 wino44_wgrad -- python3 tools/one_wgrad.py 320 320 8 > gpurun_out/r05_sq_wg_320_320_8.txt 2>&1
cat gpurun_out/r05_sq_wg_192_64_64.txt gpurun_out/r05_sq_wg_320_320_8.txt
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-sampler > gpurun_out/r05_bench1.json 2> gpurun_out/r05_bench1.err
