#!/usr/bin/env python3
"""MFMA-busy share per kernel from one rocprofv3 PMC pass:
    rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d <dir> -- python3 bench.py ...
MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 256 CUs x 4 SIMDs): share of the SIMD-cycles of the dispatch
in which the matrix pipe is busy.
usage: tools/pmc_mfma_busy.py <counter_collection.csv> <out.md> [title]"""
import collections
import csv
import re
import sys


def main():
    path, out = sys.argv[1], sys.argv[2]
    title = sys.argv[3] if len(sys.argv) > 3 else 'MFMA busy per kernel (PMC)'
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.Counter()
    for r in csv.DictReader(open(path)):
        n = re.sub(r'\(anonymous namespace\)::', '', r['Kernel_Name'])
        n = re.sub(r'\(.*', '', n)[:72]
        agg[n][r['Counter_Name']] += float(r['Counter_Value'])
        if r['Counter_Name'] == 'GRBM_GUI_ACTIVE':
            cnt[n] += 1
    tot_active = sum(v['GRBM_GUI_ACTIVE'] for v in agg.values())
    tot_mfma = sum(v['SQ_VALU_MFMA_BUSY_CYCLES'] for v in agg.values())
    lines = [f'# {title}', '',
             'rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE (own pass, weight gradients on the '
             'launch stream: FVA_WGRAD_STREAM=0, so that every dispatch has the GPU to itself).  MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / '
             '(GRBM_GUI_ACTIVE / 8 XCDs x 256 CUs x 4 SIMDs): share of the SIMD-cycles of the dispatch in which the matrix pipe is busy.', '',
             '| kernel | launches | share of GPU-active cycles % | MFMA busy % |', '|---|---:|---:|---:|']
    for n, v in sorted(agg.items(), key=lambda kv: -kv[1]['GRBM_GUI_ACTIVE']):
        share = v['GRBM_GUI_ACTIVE'] / tot_active * 100
        if share < 0.5:
            continue
        simd_cycles = v['GRBM_GUI_ACTIVE'] / 8 * 256 * 4
        lines.append(f"| `{n}` | {cnt[n]} | {share:.1f} | {v['SQ_VALU_MFMA_BUSY_CYCLES'] / simd_cycles * 100:.1f} |")
    lines += ['', f'Whole step: MFMA busy {tot_mfma / (tot_active / 8 * 256 * 4) * 100:.1f} % of SIMD-cycles.']
    open(out, 'w').write('\n'.join(lines) + '\n')
    print('\n'.join(lines))


if __name__ == '__main__':
    main()
