# A/B of the split-K plan of the weight gradients inside the whole step (bench.py, 20 steps each, one process per setting)
for cfg in "256 512" "192 512" "128 512" "128 256" "64 256" "256 256"; do
  set -- $cfg
  FVA_WGRAD_SLOTS8=$1 FVA_WGRAD_SLOTS=$2 python bench.py --steps 20 --no-cpu-baseline --no-graph 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('slots8=$1 slots=$2', d['value'], d['ms_per_step'], d['side_stream_check_ms_per_step'], d['kernels']['conv_wgrad'])"
done
