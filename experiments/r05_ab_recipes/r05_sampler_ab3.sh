# sampler A/B: GroupNorm without a launch between one-launch convs (VF_GN_LAZY), parity first
cd "$(dirname "$0")/../.." || exit 1
python -m pytest tests/test_gpu_kernels.py -q -x -k "groupnorm_without_a_launch or folded_residual or conv_small" -p no:cacheprovider 2>&1 | tail -15
python -m pytest tests/test_gpu_model.py -q -x -k "small_unet_sampler_vs_oracle or generate_chain or c1_small_unet_chain or sampler_drivers or unet_small_forward or p_mean_variance or other_geometries" -p no:cacheprovider 2>&1 | tail -15
for v in 0 1 0 1; do echo "VF_GN_LAZY=$v"; VF_GN_LAZY=$v python tools/bench_sampler.py 2>/dev/null | grep '"graph": true' | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('  B%d N%d: %.3f ms/step' % (d['batch'], d['views'], d['ms_per_step']))
"; done
python -c "
import torch, sys
sys.path.insert(0, '.')
from view_fusion_amd import sampling_bench, train
m = train.build_model(device='cuda:0', phase='test')
for N in (1, 6, 12):
    sampling_bench.time_sampler(1, N, steps=20, model=m, use_graph=True)
    print('launcher calls per step at N=%d:' % N, sampling_bench.LAST_LAUNCHES_PER_STEP)
"
