#!/bin/bash
# round-2 GPU call 3: fork of the small levels, LDS-DMA staging variants, quadtree threads; kernel trace of the default
set -e
o=gpurun_out
mkdir -p $o
export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q > $o/r02_t3.log 2>&1 || { tail -40 $o/r02_t3.log; exit 1; }
tail -2 $o/r02_t3.log
ORBX_LIB=$PWD/tools/bin/liborbx_glds2.so python -m pytest tests/test_gpu_parity.py -m gpu -x -q > $o/r02_t3_glds2.log 2>&1 || { tail -40 $o/r02_t3_glds2.log; exit 1; }
tail -2 $o/r02_t3_glds2.log
b() { tag=$1; shift; env "$@" python bench.py --no-cpu-baseline --stages > $o/r02_b3_$tag.log 2>&1; echo "== $tag"; grep -E "k_pyr|k_fast|k_quad|k_desc|k_match" $o/r02_b3_$tag.log | tr -s ' ' | cut -d' ' -f2,3 | tr '\n' ' '; echo; tail -1 $o/r02_b3_$tag.log | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'])"; }
b fork X=1
b nofork ORBX_FORK_LEVEL=0
b fork3 ORBX_FORK_LEVEL=3
b fork5 ORBX_FORK_LEVEL=5
for v in glds1 glds2 glds2g4 g4 qt256; do b $v ORBX_LIB=$PWD/tools/bin/liborbx_$v.so; done
b fork_s3 X=1
rocprofv3 --kernel-trace --output-format csv -d $o/r02_trace3 -- python3 bench.py --no-cpu-baseline --steps 4 --warmup 2 > $o/r02_trace3.log 2>&1
ls $o/r02_trace3/*/ | head
