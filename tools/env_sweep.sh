#!/bin/bash
# tools/env_sweep.sh KERNEL VAR value...
k=$1; var=$2; shift; shift
for v in "$@"; do
  env $var=$v python bench.py --no-cpu-baseline --steps 8 --warmup 2 2>/dev/null | python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        j = json.loads(line); print('$var=$v', '$k', j['kernel_ms_per_step']['$k'], 'fps', j['value'])
"
done
