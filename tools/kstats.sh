#!/bin/bash
# per-kernel totals of the eager step under rocprofv3:  tools/kstats.sh <out dir> [ENV=VAL ...]   (5 timed steps + 3 warm-up)
out=$1; shift
mkdir -p "$out"
for kv in "$@"; do export "$kv"; done
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d "$out/prof" -o run -- python3 bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-graph --no-secondary > "$out/bench.json" 2> "$out/bench.err"
python3 - "$out" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/prof/**/*kernel_stats.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: -float(r['TotalDurationNs']))
tot = sum(float(r['TotalDurationNs']) for r in rows)
with open(sys.argv[1] + '/kernel_stats.txt', 'w') as o:
    o.write('total GPU kernel time %.3f ms over 8 steps (+ set-up)\n' % (tot / 1e6))
    for r in rows[:45]:
        o.write('%9.3f ms %6d calls %9.1f us avg  %s\n' % (float(r['TotalDurationNs']) / 1e6, int(r['Calls']), float(r['AverageNs']) / 1e3, r['Name'][:150]))
PY
