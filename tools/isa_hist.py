#!/usr/bin/env python3
"""Instruction histogram of one kernel from a hipcc -save-temps gfx950 .s file.
usage: isa_hist.py file.s <regex on mangled name> [--dump]"""
import re, sys
from collections import Counter
s = open(sys.argv[1]).read()
pat = sys.argv[2]
starts = [(m.start(), m.group(1)) for m in re.finditer(r'^(_Z\w+):', s, re.M)]
for idx, (pos, name) in enumerate(starts):
    if not re.search(pat, name):
        continue
    end = starts[idx + 1][0] if idx + 1 < len(starts) else len(s)
    body = s[pos:end].split('s_endpgm')[0]
    lines = [l.strip() for l in body.splitlines()[1:]]
    ins = [l for l in lines if l and not l.startswith(('.', ';', '//')) and not l.endswith(':')]
    ops = Counter(l.split()[0] for l in ins)
    print(name)
    print(len(ins), 'instructions')
    print('  '.join(f'{k}:{v}' for k, v in ops.most_common(50)))
    if '--dump' in sys.argv:
        print('\n'.join(lines))
