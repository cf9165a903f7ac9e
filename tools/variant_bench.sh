#!/bin/bash
# default bench with variant libraries (tools/build_variant.sh NAME -D...): bash tools/variant_bench.sh NAME...
run() { ORBX_LIB=$1 python bench.py --no-cpu-baseline --steps 10 2>/dev/null | python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        j = json.loads(line); print('${2:-shipped}', {k: v for k, v in j['kernel_ms_per_step'].items() if v}, 'ms/step', j['ms_per_step'], 'value', j['value'])
"; }
run "" shipped
for v in "$@"; do run $PWD/tools/bin/liborbx_$v.so $v; done
run "" shipped
