# sampler A/B: graph side branches on / off (VF_GRAPH_BRANCHES), parity first
cd "$(dirname "$0")/../.." || exit 1
python -m pytest tests/test_gpu_model.py -q -x -k "small_unet_sampler_vs_oracle or generate_chain or c1_small_unet_chain" -p no:cacheprovider 2>&1 | tail -3
for v in 0 1 0 1; do echo "VF_GRAPH_BRANCHES=$v"; VF_GRAPH_BRANCHES=$v python tools/bench_sampler.py 2>/dev/null | grep '"graph": true' | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('  B%d N%d: %.3f ms/step' % (d['batch'], d['views'], d['ms_per_step']))
"; done
