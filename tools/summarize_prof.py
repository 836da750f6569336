#!/usr/bin/env python3
"""Turn a rocprofv3 --kernel-trace --stats CSV into the per-kernel summary kept under profiles/.
usage: tools/summarize_prof.py <kernel_stats.csv> <steps_profiled | 0> [title]
steps 0 = count them: the optimizer kernel runs once per step (bench.py also steps while it settles, untimed)."""
import csv
import re
import subprocess
import sys


def demangle(n):
    if n.startswith('_Z'):
        try:
            n = subprocess.run(['c++filt', n], capture_output=True, text=True).stdout.strip() or n
        except OSError:
            pass
    n = re.sub(r'\(anonymous namespace\)::', '', n)
    return re.sub(r'\(.*$', '', n)[:72]


def main():
    path, steps = sys.argv[1], int(sys.argv[2])
    title = sys.argv[3] if len(sys.argv) > 3 else path
    rows = list(csv.DictReader(open(path)))
    if steps == 0:
        steps = max([int(r['Calls']) for r in rows if 'adam_kernel' in r['Name']] or [1])
    tot = sum(float(r['TotalDurationNs']) for r in rows)
    print(f'# {title}\n')
    print(f'rocprofv3 --kernel-trace --stats; {steps} steps profiled (warm-up included); total kernel time '
          f'{tot / 1e6 / steps:.2f} ms/step\n')
    print('| kernel | ms/step | calls/step | avg us | % |')
    print('|---|---:|---:|---:|---:|')
    for r in rows:
        pct = float(r['Percentage'])
        if pct < 0.05:
            continue
        print(f"| `{demangle(r['Name'])}` | {float(r['TotalDurationNs']) / 1e6 / steps:.3f} | {int(r['Calls']) / steps:.1f} | "
              f"{float(r['AverageNs']) / 1e3:.1f} | {pct:.1f} |")


if __name__ == '__main__':
    main()
