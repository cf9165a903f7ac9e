#!/bin/bash
# After `bash tools/r02_profiles.sh TAG` on the GPU: turn gpurun_out/r02_TAG_* into the committed profiles/r02_* artefacts.
set -e
t=${1:?tag}
o=gpurun_out
python tools/make_traffic_json.py $o/r02_${t}_pmc_fetch $o/r02_${t}_pmc_write profiles/r02_traffic.json 1024 640 480 1000 > /dev/null
python tools/make_sq_json.py $o/r02_${t}_pmc_sq profiles/r02_sq.json 1024 640 480 1000 > /dev/null
cp $o/r02_${t}_pmc_sq.log profiles/r02_sq_counters_per_kernel.log
for cfg in "kitti_stereo 256 1241 376 2000" "euroc_stereo 256 752 480 1200" "hd1080 128 1920 1080 4000"; do
  set -- $cfg
  python tools/make_traffic_json.py $o/r02_${t}_$1_pmc_fetch $o/r02_${t}_$1_pmc_write profiles/r02_traffic_$1.json $2 $3 $4 $5 $1 > /dev/null
  python tools/make_sq_json.py $o/r02_${t}_$1_pmc_sq profiles/r02_sq_$1.json $2 $3 $4 $5 $1 > /dev/null
done
for c in tum_fork3 tum_streams3 tum_streams3_fork3; do cp $o/r02_${t}_bench_$c.log profiles/r02_bench_$c.log; done
cp $o/r02_${t}_policy_rates.log profiles/r02_policy_rates.log
cp $o/r02_${t}_policy_rates.json profiles/r02_policy_rates.json
cp $o/r02_${t}_prof_tum/*/*_kernel_stats.csv profiles/r02_tum_kernel_stats.csv
cp $o/r02_${t}_prof_kitti/*/*_kernel_stats.csv profiles/r02_kitti_stereo_kernel_stats.csv
python - <<PY
import json, csv
from orb_slam2_detailed_comments_amd import build
t = json.load(open('profiles/r02_traffic.json'))
print('hash', build.kernels_hash(), t['kernels_sha256_16'], 'HBM bytes/frame', t['hbm_bytes_per_frame'])
print({k: round(v['hbm_bytes_per_launch'] / 1e6, 1) for k, v in t['kernels'].items()})
print({r['Name'][:16]: round(float(r['AverageNs']) / 1e3, 1) for r in list(csv.DictReader(open('profiles/r02_tum_kernel_stats.csv')))[:6]})
PY
# the bench lines with the counters filled in need one more GPU call (the logs of the profiling run predate the JSON files):
echo "next: gpurun 'bash tools/r02_final_logs.sh $t' and copy gpurun_out/r02_${t}_final_*.log to profiles/"
