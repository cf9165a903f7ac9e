#!/bin/bash
set -e
o=gpurun_out
python -m pytest tests/test_gpu_parity.py -m gpu -x -q > $o/r02_t10.log 2>&1 || { tail -40 $o/r02_t10.log; exit 1; }
tail -2 $o/r02_t10.log
b() { t=$1; shift; env "$@" python bench.py --no-cpu-baseline --stages > $o/r02_b10_$t.log 2>&1; echo "== $t"; grep -E "k_pyr|k_fast|k_quad|k_desc|k_match" $o/r02_b10_$t.log | tr -s ' ' | cut -d' ' -f2,3 | tr '\n' ' '; echo; tail -1 $o/r02_b10_$t.log | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'])"; }
b vt1 X=1
b vt0 ORBX_LIB=$PWD/tools/bin/liborbx_vt0.so
b vt1b X=1
