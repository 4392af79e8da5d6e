#!/usr/bin/env python3
"""Register / scratch / LDS use of the kernels in a `hipcc -save-temps` gfx950 .s file (from its amdhsa metadata), with the
waves per SIMD the VGPR count allows (512 registers per lane and SIMD, allocation granule 8, at most 8 waves).
usage: kernel_regs.py file.s [regex on the demangled-ish name]"""
import re
import sys

text = open(sys.argv[1]).read()
pat = re.compile(sys.argv[2]) if len(sys.argv) > 2 else None
meta = text[text.index('amdhsa.kernels:'):]
for blk in re.split(r'\n  - \.agpr_count:', meta)[1:]:
    blk = '.agpr_count:' + blk
    get = lambda k: (re.search(r'\.%s:\s+(\S+)' % k, blk) or [None, '?'])[1]
    name = get('name')
    if pat and not pat.search(name):
        continue
    v, a = int(get('vgpr_count')), int(get('agpr_count'))
    total = v                     # (.vgpr_count is the unified count on gfx90a+: arch VGPRs + AGPRs)
    waves = min(8, 512 // max(8, (total + 7) // 8 * 8))
    print(f"{name[:110]:110s} vgpr {v:3d} agpr {a:3d} sgpr {get('sgpr_count'):>3s} spill v{get('vgpr_spill_count')} s{get('sgpr_spill_count')} "
          f"scratch {get('private_segment_fixed_size'):>5s} lds {get('group_segment_fixed_size'):>6s}  waves/SIMD {waves}")
