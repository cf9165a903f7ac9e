#!/bin/bash
# Everything profiles/r02_* is refreshed from, in one GPU call:  bash tools/r02_profiles.sh TAG   (outputs under gpurun_out/)
set -e
tag=${1:-a}
export TMPDIR=/tmp
o=gpurun_out
mkdir -p $o
python -m pytest tests -m gpu -x -q > $o/r02_${tag}_tests.log 2>&1 || { tail -60 $o/r02_${tag}_tests.log; exit 1; }
tail -2 $o/r02_${tag}_tests.log
python tools/policy_rates.py --json $o/r02_${tag}_policy_rates.json > $o/r02_${tag}_policy_rates.log 2>&1 || { tail -30 $o/r02_${tag}_policy_rates.log; exit 1; }
cat $o/r02_${tag}_policy_rates.log
python bench.py --stages > $o/r02_${tag}_bench_tum.log 2>&1; tail -1 $o/r02_${tag}_bench_tum.log | cut -c1-300
python bench.py --fork-level 3 --no-cpu-baseline > $o/r02_${tag}_bench_tum_fork3.log 2>&1; tail -1 $o/r02_${tag}_bench_tum_fork3.log | cut -c1-200
python bench.py --streams 3 --no-cpu-baseline > $o/r02_${tag}_bench_tum_streams3.log 2>&1; tail -1 $o/r02_${tag}_bench_tum_streams3.log | cut -c1-200
python bench.py --streams 3 --fork-level 3 --no-cpu-baseline > $o/r02_${tag}_bench_tum_streams3_fork3.log 2>&1; tail -1 $o/r02_${tag}_bench_tum_streams3_fork3.log | cut -c1-200
for c in kitti_stereo euroc_stereo hd1080; do
  python bench.py --config $c --stages > $o/r02_${tag}_bench_$c.log 2>&1; tail -1 $o/r02_${tag}_bench_$c.log | cut -c1-300
done
rocprofv3 --kernel-trace --stats --output-format csv -d $o/r02_${tag}_prof_tum -- python3 bench.py --no-cpu-baseline > $o/r02_${tag}_prof_tum.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $o/r02_${tag}_prof_kitti -- python3 bench.py --config kitti_stereo --no-cpu-baseline > $o/r02_${tag}_prof_kitti.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $o/r02_${tag}_pmc_fetch -- python3 tools/pmc_probe.py 1024 > $o/r02_${tag}_pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $o/r02_${tag}_pmc_write -- python3 tools/pmc_probe.py 1024 > $o/r02_${tag}_pmc_write.log 2>&1
bash tools/pmc_passes.sh $o/r02_${tag}_pmc_sq > $o/r02_${tag}_pmc_sq.log 2>&1
bash tools/r02_profiles_configs.sh $tag
echo profiles done
