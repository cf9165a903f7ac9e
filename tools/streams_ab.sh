#!/bin/bash
run() { label=$1; shift; lib=$1; shift; if [ "$lib" != "-" ]; then export ORBX_LIB=$lib; else unset ORBX_LIB; fi
  timeout -k 10 200 python bench.py --no-cpu-baseline --no-host-io --steps 12 --warmup 3 "$@" 2>/dev/null | python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        j = json.loads(line); print('$label', '$*', 'fps', j['value'], 'step', j['ms_per_step'])
"; }
for rep in 1 2; do
run cur - 
run cur - --streams 3 --fork-level 3
run wps4 tools/bin/liborbx_wps4.so
run wps4 tools/bin/liborbx_wps4.so --streams 3 --fork-level 3
run wps4 tools/bin/liborbx_wps4.so --streams 2
run wps4 tools/bin/liborbx_wps4.so --streams 4 --fork-level 3
run cur - --streams 4 --fork-level 3
done
