cd "$(dirname "$0")/../.." || exit 1
for v in 0 1 0 1; do VF_CONV1X1_64=$v python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-sampler 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline']['kernels']['direct_conv']
print('VF_CONV1X1_64=$v  step %.3f ms   direct_conv (eager event table) %.3f ms frac %.3f' % (d['ms_per_step'], k['ms_per_step'], k['frac_of_fp32_mfma_peak']))"; done
