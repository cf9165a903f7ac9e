#!/bin/bash
# quadtree workgroup size against the launch shape: bash tools/qt_threads_sweep.sh CONFIG...
for c in "$@"; do for t in 256 512 1024; do
  ORBX_QT_THREADS=$t python bench.py --config $c --no-cpu-baseline --steps 10 2>/dev/null | python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        j = json.loads(line); print('$c', 'threads', $t, 'k_quadtree ms/step', j['kernel_ms_per_step']['k_quadtree'], 'value', j['value'])
"
done; done
