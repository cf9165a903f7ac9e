#!/bin/bash
run() { # label, env...
  label=$1; shift
  env "$@" python bench.py --no-cpu-baseline --streams 1 --steps 8 --warmup 2 2>/dev/null | python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        j = json.loads(line); print('$label', 'k_fast', j['kernel_ms_per_step']['k_fast_rows'], 'fps', j['value'])
"
}
for v in "$@"; do run $v ORBX_LIB=$PWD/tools/bin/liborbx_$v.so; done
