#!/bin/bash
# frames/s against the batch size per configuration: bash tools/batch_sweep.sh
run() { python bench.py --config $1 --batch $2 --no-cpu-baseline --steps 8 --warmup 2 2>/dev/null | python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        j = json.loads(line); print('$1', 'batch', $2, 'ms/step', j['ms_per_step'], 'value', j['value'], 'fast frac', j['roofline']['frac'])
"; }
for b in 128 256 512 1024; do run tum $b; done
for b in 32 64 128; do run hd1080 $b; done
for b in 64 128 256; do run kitti_stereo $b; done
for b in 128 256; do run euroc_stereo $b; done
