# backward apply pass as a persistent grid (FVA_APPLY_PERSIST = blocks per CU): does the side stream's wgrad get in beside it?
mkdir -p gpurun_out/persist
for k in 0 2 3 4 0; do
  for w8 in 1 0; do
    FVA_APPLY_PERSIST=$k FVA_WGRAD8=$w8 python bench.py --steps 30 --no-secondary --no-cpu-baseline 2>/dev/null > gpurun_out/persist/p${k}_w${w8}.json
    python - <<PY
import json
d = json.loads(open('gpurun_out/persist/p${k}_w${w8}.json').read().strip().splitlines()[-1])
print('persist', $k, 'wgrad8', $w8, d['ms_per_step'], d['side_stream_check_ms_per_step'])
PY
  done
done
