#!/bin/bash
# tools/stress_suite.sh [NPROC] [REPEATS]: the whole GPU test-suite in NPROC (<= 2: some tests start a child process, and a box allows 6 GPU processes) processes at once, REPEATS times each --
# ordering bugs between streams show up under a loaded GPU, not on an idle one
np=${1:-2}; reps=${2:-2}
pids=()
for i in $(seq 1 $np); do
  ( for r in $(seq 1 $reps); do python -m pytest tests -m gpu -x -q -p no:cacheprovider > gpurun_out/stress_${i}_$r.log 2>&1 || exit 1; done ) &
  pids+=($!)
done
rc=0
for p in "${pids[@]}"; do wait $p || rc=1; done
tail -qn 1 gpurun_out/stress_*.log
exit $rc
