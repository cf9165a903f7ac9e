#!/bin/bash
# PMC counters of the non-headline configurations, taken on bench.py itself (every dispatch of a kernel in the run is the same
# workload: warm-up, calibration and timed steps):  bash tools/r03_profiles_configs.sh TAG  -> gpurun_out/r03_TAG_<config>_pmc_*
set -e
tag=${1:-a}
export TMPDIR=/tmp
o=gpurun_out
for c in ${CONFIGS:-kitti_stereo euroc_stereo hd1080}; do
  cmd="python3 bench.py --config $c --no-cpu-baseline --no-host-io --steps 4 --warmup 1"   # (no host_io: its 64-frame chunks would be averaged into the per-launch counters)
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $o/r03_${tag}_${c}_pmc_fetch -- $cmd > $o/r03_${tag}_${c}_pmc_fetch.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $o/r03_${tag}_${c}_pmc_write -- $cmd > $o/r03_${tag}_${c}_pmc_write.log 2>&1
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD --kernel-trace --output-format csv -d $o/r03_${tag}_${c}_pmc_sq/p1 -- $cmd > $o/r03_${tag}_${c}_pmc_sq.log 2>&1
  rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $o/r03_${tag}_${c}_pmc_sq/p4 -- $cmd >> $o/r03_${tag}_${c}_pmc_sq.log 2>&1
  rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $o/r03_${tag}_${c}_pmc_sq/p5 -- $cmd >> $o/r03_${tag}_${c}_pmc_sq.log 2>&1
  echo "$c counters done"
done
