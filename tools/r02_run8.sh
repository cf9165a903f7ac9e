#!/bin/bash
set -e
o=gpurun_out
b() { t=$1; shift; env "$@" python bench.py --no-cpu-baseline --stages > $o/r02_b8_$t.log 2>&1; echo "== $t"; grep -E "k_pyr|k_fast|k_quad|k_desc|k_match" $o/r02_b8_$t.log | tr -s ' ' | cut -d' ' -f2,3 | tr '\n' ' '; echo; tail -1 $o/r02_b8_$t.log | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'])"; }
b default X=1
b rrhalf ORBX_LIB=$PWD/tools/bin/liborbx_rrhalf.so
b default2 X=1
