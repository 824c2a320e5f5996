#!/usr/bin/env python
"""Per-kernel register / scratch report of libsslrec_hip.so's sources (hipcc -Rpass-analysis=kernel-resource-usage).
usage: python tools/resource_usage.py [out.json]   (no GPU needed)"""
import json, os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'sslrec_amd', 'csrc')
rows = []
for src in ('spmm.hip', 'spmm_swept.hip', 'losses.hip', 'infonce.hip', 'eval.hip', 'mt19937.hip'):
    if not os.path.exists(os.path.join(CSRC, src)):
        continue
    p = subprocess.run(['hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-c', src, '-o', '/dev/null',
                        '-Rpass-analysis=kernel-resource-usage'], cwd=CSRC, capture_output=True, text=True)
    cur = None
    for line in p.stderr.splitlines():
        m = re.search(r'remark: (?:\s*)([A-Za-z ]+?)(?: \[bytes/lane\]| \[waves/SIMD\]| \[bytes/block\])?: (\S+)', line)
        if not m:
            continue
        key, val = m.group(1).strip(), m.group(2)
        if key == 'Function Name':
            name = subprocess.run(['c++filt', val], capture_output=True, text=True).stdout.strip() or val
            cur = {'source': src, 'kernel': name.replace('void ', '').split('(')[0]}
            rows.append(cur)
        elif cur is not None and key in ('VGPRs', 'AGPRs', 'ScratchSize', 'Occupancy', 'SGPRs', 'LDS Size', 'VGPRs Spill', 'SGPRs Spill'):
            cur[key] = int(val)
spill = [r for r in rows if r.get('ScratchSize', 0) or r.get('VGPRs Spill', 0)]
for r in rows:
    print('%-14s %-70s VGPR %3d AGPR %3d scratch %4d occ %d' % (r['source'], r['kernel'][:70], r.get('VGPRs', 0), r.get('AGPRs', 0),
                                                               r.get('ScratchSize', 0), r.get('Occupancy', 0)))
print('%d kernels, %d with scratch' % (len(rows), len(spill)))
if len(sys.argv) > 1:
    json.dump({'command': 'hipcc --offload-arch=gfx950 -O3 -Rpass-analysis=kernel-resource-usage', 'kernels': rows,
               'kernels_with_scratch': [r['kernel'] for r in spill]}, open(sys.argv[1], 'w'), indent=1)
