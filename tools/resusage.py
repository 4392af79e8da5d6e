#!/usr/bin/env python3
"""Print VGPR / scratch / occupancy per kernel of one .hip file (hipcc -Rpass-analysis)."""
import re, subprocess, sys
src = sys.argv[1]
only_bad = len(sys.argv) > 2 and sys.argv[2] == '--bad'
out = subprocess.run(['/opt/rocm/bin/hipcc', '-O3', '-std=c++17', '--offload-arch=gfx950', '-c', src,
                      '-o', '/dev/null', '-Rpass-analysis=kernel-resource-usage'],
                     capture_output=True, text=True).stderr
cur = {}
rows = []
for line in out.splitlines():
    m = re.search(r'Function Name: (\S+)', line)
    if m:
        cur = {'name': m.group(1)}; rows.append(cur); continue
    for key, pat in (('vgpr', r'\bVGPRs: (\d+)'), ('sgpr', r'SGPRs: (\d+)'), ('scratch', r'ScratchSize \[bytes/lane\]: (\d+)'),
                     ('occ', r'Occupancy \[waves/SIMD\]: (\d+)'), ('lds', r'LDS Size \[bytes/block\]: (\d+)')):
        m = re.search(pat, line)
        if m and cur is not None:
            cur[key] = int(m.group(1))
demangle = subprocess.run(['c++filt'] + [r['name'] for r in rows], capture_output=True, text=True).stdout.splitlines()
for r, d in zip(rows, demangle):
    d = re.sub(r'\(.*', '', d).replace('void ffgpu::', '')
    if only_bad and r.get('scratch', 0) == 0 and r.get('occ', 8) == 8:
        continue
    print(f"{d:70s} vgpr={r.get('vgpr')} sgpr={r.get('sgpr')} scratch={r.get('scratch')} occ={r.get('occ')} lds={r.get('lds')}")
