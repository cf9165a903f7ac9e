#!/bin/bash
# The bench lines that go into profiles/ (run on the GPU after tools/r03_collect.sh wrote the counter files)
t=${1:?tag}
o=gpurun_out
python bench.py --stages > $o/r03_${t}_final_bench_tum.log 2>&1; tail -1 $o/r03_${t}_final_bench_tum.log | cut -c1-1400
ORBX_MATCH_KERNEL=valu python bench.py --no-cpu-baseline > $o/r03_${t}_final_bench_tum_match_valu.log 2>&1; tail -1 $o/r03_${t}_final_bench_tum_match_valu.log | cut -c1-140
ORBX_MATCH_KERNEL=i8 python bench.py --no-cpu-baseline > $o/r03_${t}_final_bench_tum_match_i8.log 2>&1; tail -1 $o/r03_${t}_final_bench_tum_match_i8.log | cut -c1-140
for c in kitti_stereo euroc_stereo hd1080; do
  python bench.py --config $c --stages > $o/r03_${t}_final_bench_$c.log 2>&1
  tail -1 $o/r03_${t}_final_bench_$c.log | python3 -c "
import sys, json
j = json.loads(sys.stdin.read()); r = j['roofline']; print(j['config']['name'], j['value'], j['ms_per_step'], r['frac'], r['traffic'], (r.get('ports') or {}).get('valu_frac'), j['cpu_baseline']['value'], j['config']['images_per_s'])"
done
for c in kitti_stereo euroc_stereo; do
  python bench.py --config $c --stereo-batch split --no-cpu-baseline > $o/r03_${t}_final_bench_${c}_split.log 2>&1; tail -1 $o/r03_${t}_final_bench_${c}_split.log | cut -c1-120
done
