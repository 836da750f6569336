# what each class of launches costs the WHOLE step: bench.py with that class skipped (results wrong, timing only)
for a in none wgrad apply bwd_apply dgrad "wgrad,apply,bwd_apply"; do
  FVA_BN_TICKET=0 FVA_ABLATE=$a python bench.py --steps 15 --no-cpu-baseline --no-graph 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('skip $a:', d['ms_per_step'], d['side_stream_check_ms_per_step'])"
done
