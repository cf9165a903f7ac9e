#!/bin/bash
# round-2 GPU call 1: GPU suite, then bench of the new grid orders against the round-1 orders (tools/bin/liborbx_base.so)
set -e
o=gpurun_out
mkdir -p $o
python -m pytest tests -m gpu -x -q > $o/r02_t1.log 2>&1 || { tail -40 $o/r02_t1.log; exit 1; }
tail -3 $o/r02_t1.log
python bench.py --no-cpu-baseline --stages > $o/r02_b1_new.log 2>&1
tail -12 $o/r02_b1_new.log
ORBX_LIB=$PWD/tools/bin/liborbx_base.so python bench.py --no-cpu-baseline --stages > $o/r02_b1_base.log 2>&1
tail -12 $o/r02_b1_base.log
