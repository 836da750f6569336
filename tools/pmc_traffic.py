#!/usr/bin/env python3
"""HBM traffic per launch from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs, as
MI355X_MICROARCH.md section HBM prescribes).  Units: the counters are in KiB; on gfx950 FETCH_SIZE reports exactly
half of a wide coalesced read, so it is doubled; WRITE_SIZE is exact.  (Check in this data: adam_kernel reads
16 B and writes 12 B per parameter, 61.95 M parameters -> 991 / 743 MB; measured 2 x 495 / 743 MB.)

usage: tools/pmc_traffic.py <fetch counter_collection.csv> <write counter_collection.csv> <out.json> <out.md>"""
import collections
import csv
import json
import re
import sys

def _epi(n, kernel):
    """epilogue template argument of a patch-kernel instance name (demangled or mangled), or None"""
    m = re.search(kernel + r'<\d+, \d+, \d+, (\d)>', n) or re.search(kernel + r'ILi\d+ELi\d+ELi\d+ELi(\d)EEE', n)
    return int(m.group(1)) if m else None


CLASSES = {'conv_fwd': lambda n: ('igemm_kernel' in n and re.search(r'ELi0ELi\d+EEE', n)) or 'igemm8_kernelILi0' in n or 'igemm8_kernel<0' in n
                                 or _epi(n, 'pconv_kernel') == 0,
           'conv_dgrad': lambda n: ('igemm_kernel' in n and re.search(r'ELi[14]ELi\d+EEE', n)) or 'igemm8_kernelILi1' in n or 'igemm8_kernel<1' in n
                                   or 'igemm8_kernelILi4' in n or 'igemm8_kernel<4' in n or _epi(n, 'pconv_kernel') in (1, 4) or 'pdgrad2_kernel' in n,
           'conv_wgrad': lambda n: 'wgrad_kernel' in n or 'wgrad8_kernel' in n or 'wgrad_reduce_kernel' in n or 'wgrad_reduce4_kernel' in n}


def load(path, counter):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        if r['Counter_Name'] == counter:
            a = agg[r['Kernel_Name']]
            a[0] += 1
            a[1] += float(r['Counter_Value'])
    return agg


def main():
    f, w = load(sys.argv[1], 'FETCH_SIZE'), load(sys.argv[2], 'WRITE_SIZE')
    out, lines = {}, ['| kernel | launches | HBM read MB/launch (2 x FETCH_SIZE) | HBM write MB/launch |', '|---|---:|---:|---:|']
    for n in sorted(f, key=lambda k: -(2 * f[k][1] + w.get(k, [0, 0])[1])):
        cnt, tot = f[n]
        wc, wt = w.get(n, [0, 0.0])
        rd, wr = 2 * tot / cnt * 1024 / 1e6, (wt / wc * 1024 / 1e6 if wc else 0.0)
        if rd + wr > 1.0:
            nm = re.sub(r'[(].*', '', n.replace('(anonymous namespace)::', ''))[:70]
            lines.append(f"| `{nm}` | {cnt} | {rd:.1f} | {wr:.1f} |")
    for cls, pred in CLASSES.items():
        names = [n for n in f if pred(n)]
        main_launches = sum(f[n][0] for n in names if 'reduce' not in n)
        w_launches = sum(w[n][0] for n in names if 'reduce' not in n and n in w) or 1      # the two passes may run a different number of steps
        rd = sum(2 * f[n][1] for n in names) * 1024 / main_launches
        wr = sum(w.get(n, [0, 0.0])[1] for n in names) * 1024 / w_launches
        out[cls] = {'bytes_per_launch': round(rd + wr), 'read_bytes_per_launch': round(rd),
                    'write_bytes_per_launch': round(wr), 'launches_profiled': main_launches,
                    'source': 'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), FETCH_SIZE doubled (gfx950)'}
    # step totals: every kernel's bytes, per optimizer step (adam_kernel runs once per step)
    steps = max([f[n][0] for n in f if 'adam_kernel' in n] or [1])
    wsteps = max([w[n][0] for n in w if 'adam_kernel' in n] or [steps])
    total_rd = sum(2 * f[n][1] for n in f) * 1024 / steps
    total_wr = sum(w[n][1] for n in w) * 1024 / wsteps
    bn = lambda n: n.startswith('bn_') or '_bn_' in n or 'bn_bwd' in n or 'bn_silu' in n
    bn_rd = sum(2 * f[n][1] for n in f if bn(n)) * 1024 / steps
    bn_wr = sum(w[n][1] for n in w if bn(n)) * 1024 / wsteps
    out['step'] = {'bytes_per_step': round(total_rd + total_wr), 'read_bytes_per_step': round(total_rd), 'write_bytes_per_step': round(total_wr),
                   'batchnorm_pass_bytes_per_step': round(bn_rd + bn_wr), 'steps_profiled': steps}
    json.dump(out, open(sys.argv[3], 'w'), indent=1)
    open(sys.argv[4], 'w').write('# HBM traffic per launch (PMC), bench.py B=32 640x640 bf16, weight gradients on the launch stream (FVA_WGRAD_STREAM=0)\n\n' + '\n'.join(lines) +
                                 '\n\nPer conv class (all tile variants pooled, per conv call):\n\n```\n' + json.dumps(out, indent=1) + '\n```\n')
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
