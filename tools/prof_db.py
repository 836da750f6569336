#!/usr/bin/env python3
"""Per-kernel summary of a rocprofv3 (rocpd sqlite) --kernel-trace result: ms per step, calls per step, average duration.
usage: tools/prof_db.py <results.db> [steps | 0 = count adam_kernel launches] [title]"""
import re
import sqlite3
import subprocess
import sys


def demangle(n):
    if n.startswith('_Z'):
        try:
            n = subprocess.run(['c++filt', n], capture_output=True, text=True).stdout.strip() or n
        except OSError:
            pass
    n = re.sub(r'\(anonymous namespace\)::', '', n)
    n = re.sub(r'^void ', '', n)
    return re.sub(r'\(.*$', '', n)[:80]


def load(path):
    db = sqlite3.connect(path)
    cols = [r[1] for r in db.execute('pragma table_info(kernels)')]
    namec = 'name' if 'name' in cols else [c for c in cols if 'name' in c][0]
    durc = 'duration' if 'duration' in cols else None
    q = f'select {namec}, count(*), sum({durc}) from kernels group by {namec}' if durc else \
        f'select {namec}, count(*), sum(end - start) from kernels group by {namec}'
    return [(demangle(n), c, t) for n, c, t in db.execute(q)]


def main():
    path = sys.argv[1]
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    title = sys.argv[3] if len(sys.argv) > 3 else path
    rows = load(path)
    if steps == 0:
        steps = max([c for n, c, t in rows if 'adam_kernel' in n] or [1])
    tot = sum(t for _, _, t in rows)
    print(f'# {title}\n')
    print(f'rocprofv3 --kernel-trace --stats; {steps} steps profiled (warm-up included); total kernel time {tot / 1e6 / steps:.2f} ms/step\n')
    print('| kernel | ms/step | calls/step | avg us | % |')
    print('|---|---:|---:|---:|---:|')
    for n, c, t in sorted(rows, key=lambda r: -r[2]):
        if t / tot < 0.0005:
            continue
        print(f'| `{n}` | {t / 1e6 / steps:.3f} | {c / steps:.1f} | {t / c / 1e3:.1f} | {100 * t / tot:.1f} |')


if __name__ == '__main__':
    main()
