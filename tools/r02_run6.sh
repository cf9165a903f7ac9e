#!/bin/bash
# round-2 GPU call 6: FAST appends its survivors to the dense per-level arrays (no quadtree gather)
set -e
o=gpurun_out
mkdir -p $o
export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q > $o/r02_t6.log 2>&1 || { tail -60 $o/r02_t6.log; exit 1; }
tail -2 $o/r02_t6.log
b() { tag=$1; shift; env "$@" python bench.py --no-cpu-baseline --stages > $o/r02_b6_$tag.log 2>&1; echo "== $tag"; grep -E "k_pyr|k_fast|k_quad|k_desc|k_match" $o/r02_b6_$tag.log | tr -s ' ' | cut -d' ' -f2,3 | tr '\n' ' '; echo; tail -1 $o/r02_b6_$tag.log | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'])"; }
b default X=1
b fork3 ORBX_FORK_LEVEL=3
b default2 X=1
b keys1400 ORBX_QT_LDS_KEYS=1400
