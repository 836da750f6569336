#!/usr/bin/env python3
"""One training step of a rocprofv3 --kernel-trace result (rocpd sqlite), grouped by (kernel, blocks, stream): count, total and
average duration -- the per-launch-shape view that the per-kernel summary of tools/prof_db.py averages away.
usage: tools/step_kernels.py <results.db> [rows=60] [filter substring]"""
import collections
import re
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    top = int(sys.argv[2]) if len(sys.argv) > 2 else 60
    flt = sys.argv[3] if len(sys.argv) > 3 else ''
    rows = list(db.execute('select name, start, end, grid_x, grid_y, workgroup_x, stream_id from kernels order by start'))
    marks = [i for i, r in enumerate(rows) if 'adam_kernel' in r[0]]
    if len(marks) < 2:
        raise SystemExit('fewer than two optimizer steps in the trace')
    step = rows[marks[-2] + 1:marks[-1] + 1]
    agg = collections.defaultdict(lambda: [0, 0.0])
    for n, s, e, gx, gy, wx, st in step:
        n = re.sub(r'\(anonymous namespace\)::', '', n)
        n = re.sub(r'^void ', '', n)[:56]
        k = (n, gx // max(wx, 1), gy, st)
        agg[k][0] += 1
        agg[k][1] += (e - s) / 1e3
    tot = sum(v[1] for v in agg.values())
    print(f'{len(step)} kernels in the last step, wall {(step[-1][2] - step[0][1]) / 1e6:.2f} ms, summed kernel time {tot / 1e3:.2f} ms')
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        if flt and flt not in k[0]:
            continue
        if top <= 0:
            break
        top -= 1
        print(f'{v[1]:8.1f} us  n={v[0]:3d} avg {v[1] / v[0]:7.1f}  {k[0]}  blocks={k[1]}x{k[2]} stream={k[3]}')


if __name__ == '__main__':
    main()
