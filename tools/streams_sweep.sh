#!/bin/bash
# frames/s against the number of extract+match pipelines per GPU
for n in 1 2 3 4 5 6; do
  python bench.py --no-cpu-baseline --streams $n 2>/dev/null | python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        j = json.loads(line); print('streams', $n, 'fps', j['value'], 'ms/step', j['ms_per_step'])
"
done
