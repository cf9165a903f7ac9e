#!/bin/bash
# Phase timing needs the timing-knob build (the shipped library ignores ORBX_*_STOP: results are wrong with a knob set):
#   tools/build_variant.sh knobs -DORBX_TIMING_KNOBS        (-> tools/bin/liborbx_knobs.so)
test -f tools/bin/liborbx_knobs.so || bash tools/build_variant.sh knobs -DORBX_TIMING_KNOBS
export ORBX_LIB=$PWD/tools/bin/liborbx_knobs.so
# k_fast_rows time per phase: cumulative runs with ORBX_FAST_STOP (results are wrong for STOP != 0; timing only)
for s in 1 2 3 4 0; do
  ORBX_FAST_STOP=$s python bench.py --no-cpu-baseline --streams 1 --steps 8 --warmup 2 2>/dev/null | python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        j = json.loads(line); print('stop', $s, 'k_fast ms/step', j['kernel_ms_per_step']['k_fast_rows'], 'fps', j['value'])
"
done
