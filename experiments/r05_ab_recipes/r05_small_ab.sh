# A/B of two builds of the C ABI on one box.  build/ab/libvf_base*.so = the library of the commit BEFORE the change under test:
#   git stash; python -m view_fusion_amd.build; mkdir -p build/ab; cp view_fusion_amd/lib/libvf_hip.so build/ab/libvf_baseN.so; git stash pop; python -m view_fusion_amd.build
# (scratch files, removed after the round's measurements; results: profiles/r05_*.md)
cd "$(dirname "$0")/../.." || exit 1
python -m pytest tests/test_gpu_kernels.py -q -x -k "conv_small or folded_residual or groupnorm_without" -p no:cacheprovider 2>&1 | tail -3
python -m pytest tests/test_gpu_model.py -q -x -k "small_unet_sampler_vs_oracle or generate_chain or c1_small_unet_chain" -p no:cacheprovider 2>&1 | tail -3
for i in 1 2; do
echo "== base"; VF_DEBUG_AB=1 VF_HIP_LIB=$PWD/build/ab/libvf_base3.so python tools/bench_sampler.py 2>/dev/null | grep '"graph": true' | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('  B%d N%d: %.3f ms/step' % (d['batch'], d['views'], d['ms_per_step']))
"
echo "== new"; python tools/bench_sampler.py 2>/dev/null | grep '"graph": true' | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('  B%d N%d: %.3f ms/step' % (d['batch'], d['views'], d['ms_per_step']))
"
done
python tools/small_gn_bench.py 1 2>/dev/null | sed 's/  stats.*gn_launch/  gn_launch/'
