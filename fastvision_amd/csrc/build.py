#!/usr/bin/env python3
"""Build libfastvision_amd.so for gfx950 with hipcc (cross-compiles without a GPU).

    python fastvision_amd/csrc/build.py [--report] [--force]

Objects are cached under fastvision_amd/csrc/build/ by source mtime; the shared library lands next to
the sources (in-tree, git-ignored) so that it travels to the GPU box with the repo snapshot.
"""
import os
import re
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
LIB = os.path.join(HERE, 'libfastvision_amd.so')
SOURCES = ['errors.hip', 'conv_igemm.hip', 'conv_wgrad.hip', 'bn_act.hip', 'stem.hip', 'head.hip', 'loss.hip',
           'optim.hip', 'detect.hip', 'pipeline.hip', 'colour.hip', 'roi.hip', 'vgg.hip']
# -ffp-contract=off: the matcher must reproduce the reference's fp32 op order bit for bit (no FMA fusion)
FLAGS = ['--offload-arch=gfx950', '-O3', '-fPIC', '-std=c++17', '-Wall', '-Wno-unused-function'] + os.environ.get('FVA_EXTRA_FLAGS', '').split()   # experiments: e.g. -DFVA_NT_STORES=1 (use with --force)
PER_FILE = {'colour.hip': ['-ffp-contract=off'], 'loss.hip': ['-ffp-contract=off'], 'detect.hip': ['-ffp-contract=off'], 'roi.hip': ['-ffp-contract=off']}


def _newer(src, obj):
    deps = [src, os.path.join(HERE, 'common.h'), os.path.join(ROOT, 'include', 'fastvision_amd.h')]
    return (not os.path.exists(obj)) or any(os.path.getmtime(d) > os.path.getmtime(obj) for d in deps)


def compile_one(name, report, force):
    src = os.path.join(HERE, name)
    obj = os.path.join(HERE, 'build', name.replace('.hip', '.o'))
    if not force and not report and not _newer(src, obj):
        return name, 0, ''
    cmd = [HIPCC] + FLAGS + PER_FILE.get(name, []) + ['-c', src, '-o', obj]
    if report:
        cmd.append('-Rpass-analysis=kernel-resource-usage')
    p = subprocess.run(cmd, capture_output=True, text=True, cwd=HERE)
    return name, p.returncode, p.stderr


def summarize(stderr):
    rows, cur = [], {}
    for line in stderr.splitlines():
        m = re.search(r'Function Name: (\S+)', line)
        if m:
            cur = {'name': m.group(1)}
            rows.append(cur)
        for key, pat in (('vgpr', r' VGPRs: (\d+)'), ('agpr', r'AGPRs: (\d+)'), ('sgpr', r'TotalSGPRs: (\d+)'),
                         ('scratch', r'ScratchSize \[bytes/lane\]: (\d+)'), ('occ', r'Occupancy \[waves/SIMD\]: (\d+)'),
                         ('lds', r'LDS Size \[bytes/block\]: (\d+)')):
            m = re.search(pat, line)
            if m and cur is not None:
                cur[key] = int(m.group(1))
    for r in rows:
        name = subprocess.run(['c++filt', r['name']], capture_output=True, text=True).stdout.strip() or r['name']
        name = re.sub(r'\(anonymous namespace\)::', '', name).split('(')[0]
        print(f"  {name[:70]:70s} vgpr={r.get('vgpr')} agpr={r.get('agpr')} sgpr={r.get('sgpr')} "
              f"scratch={r.get('scratch')} occ={r.get('occ')} lds={r.get('lds')}")


def build(report=False, force=False, verbose=True):
    os.makedirs(os.path.join(HERE, 'build'), exist_ok=True)
    srcs = [s for s in SOURCES if os.path.exists(os.path.join(HERE, s))]
    with ThreadPoolExecutor(max_workers=4) as ex:
        results = list(ex.map(lambda n: compile_one(n, report, force), srcs))
    failed = False
    for name, rc, err in results:
        diag = [l for l in err.splitlines() if ('error' in l or 'warning' in l) and 'remark' not in l]
        if rc != 0:
            failed = True
            print(f'[build] {name}: FAILED\n' + '\n'.join(err.splitlines()[:60]), file=sys.stderr)
        elif verbose and diag:
            print(f'[build] {name}:\n' + '\n'.join(diag[:20]))
        if report and rc == 0:
            print(f'[build] {name}')
            summarize(err)
    if failed:
        raise RuntimeError('hipcc failed')
    objs = [os.path.join(HERE, 'build', s.replace('.hip', '.o')) for s in srcs]
    if force or not os.path.exists(LIB) or any(os.path.getmtime(o) > os.path.getmtime(LIB) for o in objs):
        subprocess.check_call([HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + objs, cwd=HERE)
    return LIB


if __name__ == '__main__':
    lib = build(report='--report' in sys.argv, force='--force' in sys.argv)
    print('built', lib)
