#!/bin/bash
# round-2 GPU call 2: LDS-DMA FAST staging + quadtree residency; suite, A/B benches, the other configs
set -e
o=gpurun_out
mkdir -p $o
python -m pytest tests -m gpu -x -q > $o/r02_t2.log 2>&1 || { tail -40 $o/r02_t2.log; exit 1; }
tail -3 $o/r02_t2.log
python bench.py --no-cpu-baseline --stages > $o/r02_b2_new.log 2>&1; tail -11 $o/r02_b2_new.log | cut -c1-400
ORBX_LIB=$PWD/tools/bin/liborbx_noglds.so python bench.py --no-cpu-baseline --stages > $o/r02_b2_noglds.log 2>&1; tail -11 $o/r02_b2_noglds.log | cut -c1-200
ORBX_QT_LDS_KEYS=4096 python bench.py --no-cpu-baseline --stages > $o/r02_b2_qt4096.log 2>&1; tail -11 $o/r02_b2_qt4096.log | cut -c1-200
for c in kitti_stereo euroc_stereo hd1080; do
  python bench.py --config $c --no-cpu-baseline --stages --steps 6 > $o/r02_b2_$c.log 2>&1; tail -11 $o/r02_b2_$c.log | cut -c1-700
done
