#!/bin/bash
# same-box A/B of the whole step: tools/ab_bench.sh <out dir> "<label>=<ENV=VAL ...>" ...   (eager two-stream steps, 20 timed steps each, twice)
out=$1; shift; mkdir -p "$out"
for rep in 1 2; do
  for spec in "$@"; do
    label=${spec%%=*}; envs=${spec#*=}
    env $envs python bench.py --steps 20 --no-cpu-baseline --no-graph --no-secondary 2>"$out/$label.$rep.err" > "$out/$label.$rep.json"
    python - "$out/$label.$rep.json" "$label" <<'PY' | tee -a "$out/summary.txt"
import json, sys
d = json.load(open(sys.argv[1]))
k = d['kernels']
print(sys.argv[2], d['ms_per_step'], d['side_stream_check_ms_per_step'], {c: (k[c]['ms_per_step'], k[c]['tflops']) for c in k}, 'conv3x3', d['conv3x3']['frac_of_peak'], 'clock', d['roofline']['shader_clock_mhz_under_load'])
PY
  done
done
