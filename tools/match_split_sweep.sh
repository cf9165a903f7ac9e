#!/bin/bash
# k_match time against the train-set split (ORBX_MATCH_WAVES = target waves per launch; split = ceil(target / (pairs*8)))
for t in 2048 4096 8192 16384 32768; do
  ORBX_MATCH_WAVES=$t python bench.py --no-cpu-baseline --streams 1 --steps 8 --warmup 2 2>/dev/null | python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        j = json.loads(line); print('target', $t, 'k_match ms/step', j['kernel_ms_per_step']['k_match'], 'fps', j['value'])
"
done
