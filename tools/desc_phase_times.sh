#!/bin/bash
for s in 1 2 3 0; do
  ORBX_DESC_STOP=$s python bench.py --no-cpu-baseline --streams 1 --steps 8 --warmup 2 2>/dev/null | python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        j = json.loads(line); print('stop', $s, 'k_describe ms/step', j['kernel_ms_per_step']['k_describe'])
"
done
