#!/bin/bash
# A/B of one environment knob on the bench, alternating runs in ONE process tree / on one box:
#   bash tools/ab_env.sh VF_WINO_SPLIT 0 1 [repeats]
# prints ms_per_step of every run (graph replay, no roofline / sampler / CPU legs).
VAR=$1; A=$2; B=$3; N=${4:-2}
for i in $(seq $N); do
  for v in $A $B; do
    env $VAR=$v python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-sampler --no-roofline 2>/dev/null | \
      python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$VAR=$v', round(d['ms_per_step'],3), 'ms/step')"
  done
done
