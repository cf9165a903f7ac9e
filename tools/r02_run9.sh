#!/bin/bash
set -e
o=gpurun_out
timeout -k 10 260 python tools/soak.py 200 7 > $o/r02_soak_extract.log 2>&1 || { tail -20 $o/r02_soak_extract.log; exit 1; }
tail -2 $o/r02_soak_extract.log
timeout -k 10 260 python tools/soak_policies.py 200 9 > $o/r02_soak_policies.log 2>&1 || { tail -20 $o/r02_soak_policies.log; exit 1; }
tail -3 $o/r02_soak_policies.log
