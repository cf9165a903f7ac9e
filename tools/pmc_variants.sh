#!/bin/bash
# one SQ instruction-count pass per library variant: tools/pmc_variants.sh OUT label[:lib.so] ...   (k_fast_rows rows printed)
out=$1; shift
export TMPDIR=/tmp
for spec in "$@"; do
  label=${spec%%:*}; lib=${spec#*:}
  if [ "$lib" != "$spec" ]; then export ORBX_LIB=$lib; else unset ORBX_LIB; fi
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES --kernel-trace --output-format csv -d $out/$label -- python3 tools/pmc_probe.py ${PMC_BATCH:-1024} > $out.$label.log 2>&1
  echo "== $label"; python3 tools/pmc_summarize.py $out/$label | grep -E "k_fast_rows|k_describe"
done
