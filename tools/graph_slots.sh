# does a graph branch of weight gradients that is planned for a fraction of the CUs replay faster than the single-stream graph?
for cfg in "256 512" "128 256" "64 128" "32 64"; do
  set -- $cfg
  FVA_WGRAD_SLOTS8=$1 FVA_WGRAD_SLOTS=$2 python bench.py --steps 10 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('slots8=$1 slots=$2', d['ms_per_step'], d['side_stream_check_ms_per_step'], d['hip_graph']['used'])"
done
