#!/bin/bash
# round-2 GPU call 5: device grid / gated candidate lists behind every policy; policy rates; bench
set -e
o=gpurun_out
mkdir -p $o
export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q > $o/r02_t5.log 2>&1 || { tail -60 $o/r02_t5.log; exit 1; }
tail -2 $o/r02_t5.log
python tools/policy_rates.py --json $o/r02_policy_rates.json > $o/r02_policy_rates.log 2>&1 || { tail -30 $o/r02_policy_rates.log; exit 1; }
cat $o/r02_policy_rates.log
b() { tag=$1; shift; env "$@" python bench.py --no-cpu-baseline --stages > $o/r02_b5_$tag.log 2>&1; echo "== $tag"; grep -E "k_pyr|k_fast|k_quad|k_desc|k_match" $o/r02_b5_$tag.log | tr -s ' ' | cut -d' ' -f2,3 | tr '\n' ' '; echo; tail -1 $o/r02_b5_$tag.log | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'])"; }
b default X=1
b fork3 ORBX_FORK_LEVEL=3
b fork4 ORBX_FORK_LEVEL=4
b default2 X=1
