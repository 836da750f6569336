mkdir -p gpurun_out/r4j
for n in 5 10 20; do
python bench.py --steps $n --no-cpu-baseline --no-graph --no-secondary 2>gpurun_out/r4j/s$n.err | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('steps $n', d['ms_per_step'], d['allocator'], max(d['step_ms']))" | tee -a gpurun_out/r4j/summary.txt
done
FVA_WGRAD_STREAM=0 python bench.py --steps 10 --no-cpu-baseline --no-graph --no-secondary 2>gpurun_out/r4j/noside.err | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('no side stream, steps 10', d['ms_per_step'], d['allocator'], max(d['step_ms']))" | tee -a gpurun_out/r4j/summary.txt
