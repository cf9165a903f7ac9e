#!/bin/bash
# tools/soak_parallel.sh SECONDS [NPROC]: NPROC (<= 5) soak processes with different seeds (the CPU oracle is the slow side)
secs=${1:-300}; np=${2:-5}
pids=()
for i in $(seq 1 $np); do
  python tools/soak.py $secs $((100 + i)) > gpurun_out/soak_$i.log 2>&1 &
  pids+=($!)
done
rc=0
for p in "${pids[@]}"; do wait $p || rc=1; done
tail -n 1 gpurun_out/soak_*.log
exit $rc
