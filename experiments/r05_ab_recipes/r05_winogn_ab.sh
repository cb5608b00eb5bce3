cd "$(dirname "$0")/../.." || exit 1
python -m pytest tests/test_gpu_kernels.py -q -x -k "winograd_fixup_evaluates or winograd_path" -p no:cacheprovider 2>&1 | tail -4
python -m pytest tests/test_gpu_model.py -q -x -k "small_unet_sampler_vs_oracle or generate_chain or c1_small_unet_chain or sampler_drivers or extrapolate_real" -p no:cacheprovider 2>&1 | tail -3
for v in 0 1 0 1; do echo "VF_WINO_GN=$v"; VF_WINO_GN=$v timeout 300 python tools/bench_sampler.py 2>/dev/null | grep '"graph": true' | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('  B%d N%d: %.3f ms/step' % (d['batch'], d['views'], d['ms_per_step']))
"; done
