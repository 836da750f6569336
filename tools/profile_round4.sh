#!/bin/bash
# Round-4 measurement set (run on the GPU box: tools/profile_round4.sh gpurun_out/<dir>): the default bench line, a kernel trace, three
# PMC passes summarised ON THE BOX (the raw counter csv files are 10-30 MB each and stay there).
set -o pipefail
out=$1; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
timeout -k 10 500 python3 bench.py > "$out/bench.json" 2> "$out/bench.err" || exit 1
B="python3 bench.py --no-cpu-baseline --no-secondary --steps 20 --no-graph"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d "$out/kt" -o kt -- $B > "$out/kt.json" 2> "$out/kt.log" || exit 1
db=$(ls "$out"/kt/*.db "$out"/kt/*/*.db 2>/dev/null | head -1)
python3 tools/prof_db.py "$db" 0 "Round 4: bench.py B=32 640x640 bf16, eager two-stream step (--no-graph)" > "$out/kernel_stats.md" 2> "$out/prof_db.err"
B="python3 bench.py --no-cpu-baseline --no-secondary --steps 5 --no-graph"
export FVA_WGRAD_STREAM=0
timeout -k 10 150 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$out/pmc_fetch" -o p -- $B > "$out/pmc_fetch.json" 2> "$out/pmc_fetch.log" || exit 1
timeout -k 10 150 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$out/pmc_write" -o p -- $B > "$out/pmc_write.json" 2> "$out/pmc_write.log" || exit 1
timeout -k 10 150 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$out/pmc_mfma" -o p -- $B > "$out/pmc_mfma.json" 2> "$out/pmc_mfma.log" || exit 1
f=$(ls "$out"/pmc_fetch/*counter_collection.csv "$out"/pmc_fetch/*/*counter_collection.csv 2>/dev/null | head -1)
w=$(ls "$out"/pmc_write/*counter_collection.csv "$out"/pmc_write/*/*counter_collection.csv 2>/dev/null | head -1)
m=$(ls "$out"/pmc_mfma/*counter_collection.csv "$out"/pmc_mfma/*/*counter_collection.csv 2>/dev/null | head -1)
python3 tools/pmc_traffic.py "$f" "$w" "$out/traffic.json" "$out/pmc_traffic.md" > /dev/null 2> "$out/pmc_traffic.err"
python3 tools/pmc_mfma_busy.py "$m" "$out/mfma_busy.md" "Round 4: MFMA busy per kernel (PMC)" > /dev/null 2> "$out/pmc_mfma.err"
rm -rf "$out"/pmc_fetch "$out"/pmc_write "$out"/pmc_mfma "$out"/kt
ls -la "$out"
