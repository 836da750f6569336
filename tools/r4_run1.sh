set -x
mkdir -p gpurun_out/r4a
run() { tag=$1; shift; env "$@" python bench.py --steps 15 --no-cpu-baseline --no-graph --no-secondary 2>gpurun_out/r4a/$tag.err | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$tag', d['ms_per_step'], d.get('side_stream_check_ms_per_step'), d.get('kernels'))" | tee -a gpurun_out/r4a/summary.txt; }
run base A=1
run wgrad8off FVA_WGRAD8=0
run persist3_w8off FVA_APPLY_PERSIST=3 FVA_WGRAD8=0
run persist3_lds_w8off FVA_APPLY_PERSIST=3 FVA_APPLY_LDS=32768 FVA_WGRAD8=0
run base2 A=1
