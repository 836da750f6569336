# does capping the backward apply pass's occupancy let side-stream weight gradients run under it? (whole step, eager)
run() { python bench.py --steps 15 --no-cpu-baseline --no-secondary --no-graph 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$1', d['ms_per_step'], d['side_stream_check_ms_per_step'])"; }
run "default"
FVA_APPLY_LDS=32768 run "cap 3 blocks/CU (32 KiB), U=2"
FVA_LIB_PATH=$PWD/fastvision_amd/csrc/variants/lib_u4.so FVA_APPLY_LDS=32768 run "cap 3 blocks/CU, U=4"
FVA_LIB_PATH=$PWD/fastvision_amd/csrc/variants/lib_u4.so FVA_APPLY_LDS=24576 run "cap 4 blocks/CU (24 KiB), U=4"
FVA_LIB_PATH=$PWD/fastvision_amd/csrc/variants/lib_u4.so FVA_APPLY_LDS=24576 FVA_WGRAD8=0 run "cap 4, U=4, all wgrad on the 128 kernel"
FVA_WGRAD8=0 run "default, all wgrad on the 128 kernel"
