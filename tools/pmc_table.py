#!/usr/bin/env python3
"""Per-kernel sums of every counter of one rocprofv3 PMC pass (csv), as a markdown table with a few ratios.
usage: tools/pmc_table.py <counter_collection.csv> <out.md> [title]"""
import collections
import csv
import re
import sys


def main():
    path, out = sys.argv[1], sys.argv[2]
    title = sys.argv[3] if len(sys.argv) > 3 else 'PMC counters per kernel'
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.defaultdict(collections.Counter)
    names = []
    for r in csv.DictReader(open(path)):
        n = re.sub(r'\(anonymous namespace\)::', '', r['Kernel_Name'])
        n = re.sub(r'\(.*', '', n)[:64]
        c = r['Counter_Name']
        if c not in names:
            names.append(c)
        agg[n][c] += float(r['Counter_Value'])
        cnt[n][c] += 1
    key = 'SQ_WAVE_CYCLES' if 'SQ_WAVE_CYCLES' in names else names[0]
    lines = [f'# {title}', '', '| kernel | launches | ' + ' | '.join(names) + ' |', '|---|---:|' + '---:|' * len(names)]
    for n, v in sorted(agg.items(), key=lambda kv: -kv[1][key])[:24]:
        lines.append(f'| `{n}` | {cnt[n][key]} | ' + ' | '.join(f'{v[c]:.3g}' for c in names) + ' |')
    if key == 'SQ_WAVE_CYCLES':
        lines += ['', 'Shares of SQ_WAVE_CYCLES (quad-cycles a wave is resident):', '',
                  '| kernel | ' + ' | '.join(c for c in names if c != key) + ' |', '|---|' + '---:|' * (len(names) - 1)]
        for n, v in sorted(agg.items(), key=lambda kv: -kv[1][key])[:24]:
            lines.append(f'| `{n}` | ' + ' | '.join(f'{v[c] / v[key] * 100:.1f} %' for c in names if c != key) + ' |')
    open(out, 'w').write('\n'.join(lines) + '\n')
    print('\n'.join(lines))


if __name__ == '__main__':
    main()
