#!/bin/bash
# After `bash tools/r03_profiles.sh TAG` on the GPU: turn gpurun_out/r03_TAG_* into the committed profiles/r03_* artefacts.
set -e
t=${1:?tag}
o=gpurun_out
python tools/make_traffic_json.py $o/r03_${t}_pmc_fetch $o/r03_${t}_pmc_write profiles/r03_traffic.json 1024 640 480 1000 > /dev/null
python tools/make_sq_json.py $o/r03_${t}_pmc_sq profiles/r03_sq.json 1024 640 480 1000 > /dev/null
cp $o/r03_${t}_pmc_sq.log profiles/r03_sq_counters_per_kernel.log
for cfg in "kitti_stereo 256 1241 376 2000" "euroc_stereo 256 752 480 1200" "hd1080 128 1920 1080 4000"; do
  set -- $cfg
  python tools/make_traffic_json.py $o/r03_${t}_$1_pmc_fetch $o/r03_${t}_$1_pmc_write profiles/r03_traffic_$1.json $2 $3 $4 $5 $1 > /dev/null
  python tools/make_sq_json.py $o/r03_${t}_$1_pmc_sq profiles/r03_sq_$1.json $2 $3 $4 $5 $1 > /dev/null
done
for c in tum_fork3 tum_streams3 tum_streams3_fork3 tum_batch256; do cp $o/r03_${t}_bench_$c.log profiles/r03_bench_$c.log; done
cp $o/r03_${t}_host_io_rate.log profiles/r03_host_io_rate.log
cp $o/r03_${t}_match_rate.log profiles/r03_match_rate.log
grep '^{' $o/r03_${t}_bench_two_rank_rehearsal.log > profiles/r03_bench_two_rank_rehearsal_gloo.log
cp $o/r03_${t}_policy_rates.log profiles/r03_policy_rates.log
cp $o/r03_${t}_policy_rates.json profiles/r03_policy_rates.json
cp $o/r03_${t}_prof_tum/*/*_kernel_stats.csv profiles/r03_tum_kernel_stats.csv
cp $o/r03_${t}_prof_kitti/*/*_kernel_stats.csv profiles/r03_kitti_stereo_kernel_stats.csv
python - <<PY
import json, csv
from orb_slam2_detailed_comments_amd import build
t = json.load(open('profiles/r03_traffic.json'))
print('hash', build.kernels_hash(), t['kernels_sha256_16'], 'HBM bytes/frame', t['hbm_bytes_per_frame'])
print({k: round(v['hbm_bytes_per_launch'] / 1e6, 1) for k, v in t['kernels'].items()})
print({r['Name'][:16]: round(float(r['AverageNs']) / 1e3, 1) for r in list(csv.DictReader(open('profiles/r03_tum_kernel_stats.csv')))[:6]})
PY
# the bench lines with the counters filled in need one more GPU call (the logs of the profiling run predate the JSON files):
echo "next: gpurun 'bash tools/r03_final_logs.sh $t' and copy gpurun_out/r03_${t}_final_*.log to profiles/"
