#!/bin/bash
# tools/variant_sweep.sh KERNEL variant...   (libs built by tools/build_variant.sh)
k=$1; shift
for v in "$@"; do
  ORBX_LIB=$PWD/tools/bin/liborbx_$v.so python bench.py --no-cpu-baseline --streams 1 --steps 8 --warmup 2 2>/dev/null | python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        j = json.loads(line); print('$v', '$k', j['kernel_ms_per_step']['$k'], 'fps', j['value'])
"
done
