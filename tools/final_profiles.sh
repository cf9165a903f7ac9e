#!/bin/bash
# Everything profiles/ is refreshed from, in one GPU call:  bash tools/final_profiles.sh TAG   (outputs under gpurun_out/)
set -e
tag=${1:-h}
export TMPDIR=/tmp
o=gpurun_out
rocprofv3 --kernel-trace --stats --output-format csv -d $o/prof_${tag}3 -- python3 bench.py --no-cpu-baseline --streams 3 > $o/prof_${tag}3.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $o/prof_${tag}1 -- python3 bench.py --no-cpu-baseline > $o/prof_${tag}1.log 2>&1
python bench.py --streams 3 > $o/bench_${tag}3.log 2>&1
python bench.py --stages > $o/bench_${tag}1.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $o/pmc_fetch_${tag} -- python3 tools/pmc_probe.py 256 > $o/pmc_fetch_${tag}.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $o/pmc_write_${tag} -- python3 tools/pmc_probe.py 256 > $o/pmc_write_${tag}.log 2>&1
bash tools/pmc_passes.sh $o/pmc_sq_${tag} > $o/pmc_sq_${tag}.log 2>&1
tail -1 $o/bench_${tag}3.log; tail -1 $o/bench_${tag}1.log
