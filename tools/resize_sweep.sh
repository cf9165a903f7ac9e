#!/bin/bash
for v in "$@"; do
  ORBX_RESIZE_IMPL=$v python bench.py --no-cpu-baseline --steps 8 --warmup 2 2>/dev/null | python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        j = json.loads(line); print('$v', 'k_pyr_resize', j['kernel_ms_per_step']['k_pyr_resize'], 'fps', j['value'])
"
done
