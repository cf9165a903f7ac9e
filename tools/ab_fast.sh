#!/bin/bash
# A/B of FAST kernel variants on one box: tools/ab_fast.sh "label:ENV=.. ENV=.." ...   (each run prints fps + per-kernel ms)
for spec in "$@"; do
  label=${spec%%:*}; envs=${spec#*:}
  env $envs timeout -k 10 200 python bench.py --no-cpu-baseline --steps ${AB_STEPS:-10} --warmup 3 2>/dev/null | python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        j = json.loads(line); k = j['kernel_ms_per_step']; print('$label', 'fps', j['value'], 'step', j['ms_per_step'], 'fast', k['k_fast_rows'], 'resize', k['k_pyr_resize'], 'l0', k['k_pyr_l0'], 'qt', k['k_quadtree'], 'desc', k['k_describe'], 'match', k['k_match'])
"
done
