cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_kernels.py -q -x -k "attention" -p no:cacheprovider 2>&1 | tail -3
for i in 1 2; do
echo "== base"; VF_DEBUG_AB=1 VF_HIP_LIB=$PWD/build/ab/libvf_base2.so python tools/bench_sampler.py 2>/dev/null | grep '"graph": true' | python -c "
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
