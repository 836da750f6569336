#!/bin/bash
# The measurement set behind profiles/rNN_*: the default bench line (with the CPU baseline), a kernel trace, three PMC passes
# (HBM read, HBM write, MFMA busy; counters in their own runs, weight gradients on the launch stream so that every dispatch has
# the GPU to itself).  Run on the GPU box: tools/profile_round.sh gpurun_out/<dir>; the summaries are made afterwards with
# tools/prof_db.py, tools/pmc_traffic.py, tools/pmc_mfma_busy.py.
set -o pipefail
out=$1; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
B="python3 bench.py --no-cpu-baseline --steps 20 --no-graph"
if [ -z "$PMC_ONLY" ]; then
timeout -k 10 400 python3 bench.py > "$out/bench.json" 2> "$out/bench.err" || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d "$out/kt" -o kt -- $B > "$out/kt.json" 2> "$out/kt.log" || exit 1
fi
B="python3 bench.py --no-cpu-baseline --steps 5 --no-graph"      # counter passes serialise every dispatch: five steps are plenty
export FVA_WGRAD_STREAM=0
timeout -k 10 120 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$out/pmc_fetch" -o p -- $B > "$out/pmc_fetch.json" 2> "$out/pmc_fetch.log" || exit 1
timeout -k 10 120 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$out/pmc_write" -o p -- $B > "$out/pmc_write.json" 2> "$out/pmc_write.log" || exit 1
timeout -k 10 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$out/pmc_mfma" -o p -- $B > "$out/pmc_mfma.json" 2> "$out/pmc_mfma.log" || exit 1
rm -f "$out"/pmc_*/p_kernel_trace.csv
ls -la "$out" "$out"/pmc_fetch
