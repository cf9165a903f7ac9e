for spec in "$@"; do
  label=${spec%%:*}; rest=${spec#*:}; cfg=${rest%%:*}; envs=${rest#*:}
  env $envs timeout -k 10 300 python bench.py --config $cfg --no-cpu-baseline --no-host-io --steps 12 --warmup 3 2>/dev/null | python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        j = json.loads(line); k = j['kernel_ms_per_step']; print('$label', '$cfg', 'fps', j['value'], 'step', j['ms_per_step'], 'fast', k['k_fast_rows'])
"
done
