#!/bin/bash
# tools/soak_parallel.sh SECONDS [NPROC] [SCRIPT]: NPROC (<= 5) soak processes with different seeds sharing the GPU (the CPU
# oracle is the slow side; the shared GPU is also what shook out the NULL-stream memset race)
secs=${1:-300}; np=${2:-5}; script=${3:-tools/soak.py}
pids=()
for i in $(seq 1 $np); do
  python $script $secs $((100 + i)) > gpurun_out/soak_$i.log 2>&1 &
  pids+=($!)
done
rc=0
for p in "${pids[@]}"; do wait $p || rc=1; done
tail -n 1 gpurun_out/soak_*.log
exit $rc
