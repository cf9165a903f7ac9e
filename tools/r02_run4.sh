#!/bin/bash
# round-2 GPU call 4: quadtree tail on the side stream, clear folded into level 0, device-scope events, describe keypoints/wave
set -e
o=gpurun_out
mkdir -p $o
export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q > $o/r02_t4.log 2>&1 || { tail -40 $o/r02_t4.log; exit 1; }
tail -2 $o/r02_t4.log
for v in kpw2 qt128; do
ORBX_LIB=$PWD/tools/bin/liborbx_$v.so python -m pytest tests/test_gpu_parity.py -m gpu -x -q > $o/r02_t4_$v.log 2>&1 || { tail -40 $o/r02_t4_$v.log; exit 1; }
tail -1 $o/r02_t4_$v.log
done
b() { tag=$1; shift; env "$@" python bench.py --no-cpu-baseline --stages > $o/r02_b4_$tag.log 2>&1; echo "== $tag"; grep -E "k_pyr|k_fast|k_quad|k_desc|k_match" $o/r02_b4_$tag.log | tr -s ' ' | cut -d' ' -f2,3 | tr '\n' ' '; echo; tail -1 $o/r02_b4_$tag.log | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'])"; }
b default X=1
b evflags2 ORBX_EVENT_FLAGS=2
b fork3 ORBX_FORK_LEVEL=3
b fork2 ORBX_FORK_LEVEL=2
b nofork ORBX_FORK_LEVEL=0
b keys1400 ORBX_QT_LDS_KEYS=1400
for v in kpw2 kpw4 qt128 qt512; do b $v ORBX_LIB=$PWD/tools/bin/liborbx_$v.so; done
b default2 X=1
b s3 X=1 
rocprofv3 --kernel-trace --output-format csv -d $o/r02_trace4 -- python3 bench.py --no-cpu-baseline --steps 4 --warmup 2 > $o/r02_trace4.log 2>&1
