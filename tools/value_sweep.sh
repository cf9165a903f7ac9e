#!/bin/bash
# tools/value_sweep.sh variant...: default bench (3 pipelines) value per library variant
for v in "$@"; do
  ORBX_LIB=$PWD/tools/bin/liborbx_$v.so python bench.py --no-cpu-baseline --steps 16 --warmup 4 2>/dev/null | python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        j = json.loads(line); print('$v', 'fps', j['value'], 'ms/step', j['ms_per_step'])
"
done
