#!/bin/bash
# Everything profiles/r03_* is refreshed from:  bash tools/r03_profiles.sh TAG [tests|bench|prof|all]   (outputs under gpurun_out/)
# (three stages so that each fits one gpurun call; `all` runs them in order)
set -e
tag=${1:-a}
stage=${2:-all}
export TMPDIR=/tmp
o=gpurun_out
mkdir -p $o
if [ $stage = tests ] || [ $stage = all ]; then
  python -m pytest tests -m gpu -x -q > $o/r03_${tag}_tests.log 2>&1 || { tail -60 $o/r03_${tag}_tests.log; exit 1; }
  tail -2 $o/r03_${tag}_tests.log
  python tools/policy_rates.py --json $o/r03_${tag}_policy_rates.json > $o/r03_${tag}_policy_rates.log 2>&1 || { tail -30 $o/r03_${tag}_policy_rates.log; exit 1; }
  cat $o/r03_${tag}_policy_rates.log
fi
if [ $stage = bench ] || [ $stage = all ]; then
  python bench.py --stages > $o/r03_${tag}_bench_tum.log 2>&1; tail -1 $o/r03_${tag}_bench_tum.log | cut -c1-300
  python bench.py --batch 256 --no-cpu-baseline > $o/r03_${tag}_bench_tum_batch256.log 2>&1; tail -1 $o/r03_${tag}_bench_tum_batch256.log | cut -c1-200
  python tools/host_io_rate.py > $o/r03_${tag}_host_io_rate.log 2>&1; HOST_IO_FRAMES=1024 python tools/host_io_rate.py >> $o/r03_${tag}_host_io_rate.log 2>&1; cat $o/r03_${tag}_host_io_rate.log
  python tools/match_rate.py > $o/r03_${tag}_match_rate.log 2>&1; tail -1 $o/r03_${tag}_match_rate.log
  python bench.py --fork-level 3 --no-cpu-baseline > $o/r03_${tag}_bench_tum_fork3.log 2>&1; tail -1 $o/r03_${tag}_bench_tum_fork3.log | cut -c1-200
  python bench.py --streams 3 --no-cpu-baseline > $o/r03_${tag}_bench_tum_streams3.log 2>&1; tail -1 $o/r03_${tag}_bench_tum_streams3.log | cut -c1-200
  python bench.py --streams 3 --fork-level 3 --no-cpu-baseline > $o/r03_${tag}_bench_tum_streams3_fork3.log 2>&1; tail -1 $o/r03_${tag}_bench_tum_streams3_fork3.log | cut -c1-200
  for c in kitti_stereo euroc_stereo hd1080; do
    python bench.py --config $c --stages > $o/r03_${tag}_bench_$c.log 2>&1; tail -1 $o/r03_${tag}_bench_$c.log | cut -c1-300
  done
  # the N > 1 code path rehearsed on this one-GPU box: two ranks sharing GPU 0, gloo (the 8-GPU run is the driver's)
  ORBX_BENCH_SHARE_GPU0=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 \
      bench.py --gpus 2 --backend gloo --batch 256 --steps 6 --warmup 2 --no-cpu-baseline > $o/r03_${tag}_bench_two_rank_rehearsal.log 2>&1
  grep '^{' $o/r03_${tag}_bench_two_rank_rehearsal.log | cut -c1-400
fi
if [ $stage = prof ] || [ $stage = all ]; then
  rocprofv3 --kernel-trace --stats --output-format csv -d $o/r03_${tag}_prof_tum -- python3 bench.py --no-cpu-baseline --no-host-io > $o/r03_${tag}_prof_tum.log 2>&1
  rocprofv3 --kernel-trace --stats --output-format csv -d $o/r03_${tag}_prof_kitti -- python3 bench.py --config kitti_stereo --no-cpu-baseline > $o/r03_${tag}_prof_kitti.log 2>&1
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $o/r03_${tag}_pmc_fetch -- python3 tools/pmc_probe.py 1024 > $o/r03_${tag}_pmc_fetch.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $o/r03_${tag}_pmc_write -- python3 tools/pmc_probe.py 1024 > $o/r03_${tag}_pmc_write.log 2>&1
  bash tools/pmc_passes.sh $o/r03_${tag}_pmc_sq > $o/r03_${tag}_pmc_sq.log 2>&1
  bash tools/r03_profiles_configs.sh $tag
fi
echo "profiles stage $stage done"
