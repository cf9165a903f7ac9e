#!/bin/bash
# activity / wait / clock counters per library variant: tools/pmc_variants2.sh OUT label[:lib.so] ...
out=$1; shift
export TMPDIR=/tmp
for spec in "$@"; do
  label=${spec%%:*}; lib=${spec#*:}
  if [ "$lib" != "$spec" ]; then export ORBX_LIB=$lib; else unset ORBX_LIB; fi
  i=0
  for grp in "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES" \
             "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAVES" \
             "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_SALU" \
             "GRBM_GUI_ACTIVE"; do
    i=$((i+1))
    rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $out/$label.p$i -- python3 tools/pmc_probe.py ${PMC_BATCH:-1024} > $out.$label.p$i.log 2>&1
    echo "== $label p$i"; python3 tools/pmc_summarize.py $out/$label.p$i | grep -E "k_fast_rows|k_describe"
    python3 - $out/$label.p$i <<'PY'
import csv, glob, sys, collections
d = collections.defaultdict(list)
for fn in glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(fn)):
        n = r['Kernel_Name'].split('(')[0].replace('void ', '')
        if n.startswith('k_fast_rows') or n.startswith('k_describe'):
            d[n].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for k, v in d.items():
    print('   duration_us', k, round(sum(v) / len(v), 1), 'n', len(v))
PY
  done
done
