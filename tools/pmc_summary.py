#!/usr/bin/env python3
"""Turn rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE counter_collection.csv files into per-launch HBM traffic.
Units/corrections (MI355X_MICROARCH.md, HBM section): counters are in KiB; on gfx950 FETCH_SIZE reports
half the bytes of a 16 B/lane coalesced stream -> doubled.  Both corrections are re-checked here against
k_copy16, which moves exactly 80 MB in and 80 MB out per launch.
usage: pmc_summary.py FETCH.csv WRITE.csv out.json out.md"""
import collections, csv, json, re, sys

def agg(path):
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        name = re.sub(r'\(.*', '', r['Kernel_Name']).replace('void ffgpu::', '').replace('void ', '')
        d[name].append(float(r['Counter_Value']))
    return {k: (sum(v) / len(v), len(v)) for k, v in d.items()}

f, w = agg(sys.argv[1]), agg(sys.argv[2])
alg = {  # algorithmic bytes per launch at n = 10^7 64-bit elements
    'k_copy16': (80e6, 80e6),
    'k_ew2<PM64<false, true>, 2, true>': (160e6, 80e6),
    'k_ew2<PM64<true, false>, 2, true>': (160e6, 80e6),
    'k_split<PM64<false, true>, 1, false, true, false, false>': (160e6, 240e6),
    'k_split<PM64<false, true>, 1, true, true, false, false>': (240e6, 240e6),
    'k_split<PM64<true, false>, 3, true, true, false, false>': (400e6, 560e6),
    'k_recombine<PM64<false, true>, 3, true>': (240e6, 80e6),
    'k_split<PM64<true, false>, 3, false, true, false, false>': (320e6, 560e6),
    'k_recombine<PM64<true, false>, 7, true>': (560e6, 80e6),
    'k_ew2<PM96, 2, true>': (240e6, 120e6),      # 12-byte elements, one per lane (dwordx3)
    'k_ew2<PM96, 0, true>': (240e6, 120e6),
    'k_ew2<PM128<true>, 2, true>': (320e6, 160e6),            # 16-byte elements (configs[3])
    'k_split<PM128<true>, 3, false, true, false, false>': (640e6, 1120e6),
    'k_recombine<PM128<true>, 7, true>': (1120e6, 160e6),
    'k_ew2<PM192, 2, true>': (480e6, 240e6),                 # 24-byte elements, wave-contiguous accesses (round 6)
    'k_split<PM192, 1, false, true, false, false>': (480e6, 720e6),
    'k_recombine<PM192, 3, true>': (720e6, 240e6),
}
out, lines = {}, ['| kernel | launches | FETCH_SIZE KiB | WRITE_SIZE KiB | read MB (2x FETCH) | write MB | traffic MB | algorithmic MB | traffic/alg |',
                  '|---|---|---|---|---|---|---|---|---|']
for k in sorted(f):
    if not k.startswith('k_'):
        continue
    rd = 2 * f[k][0] * 1024
    wr = w.get(k, (0, 0))[0] * 1024
    a = alg.get(k)
    tot = rd + wr
    out[k] = {'launches': f[k][1], 'FETCH_SIZE_KiB': round(f[k][0], 1), 'WRITE_SIZE_KiB': round(w.get(k, (0, 0))[0], 1),
              'read_bytes': round(rd), 'write_bytes': round(wr), 'traffic_bytes': round(tot),
              'algorithmic_bytes': int(sum(a)) if a else None}
    lines.append(f"| `{k}` | {f[k][1]} | {f[k][0]:.1f} | {w.get(k,(0,0))[0]:.1f} | {rd/1e6:.2f} | {wr/1e6:.2f} | {tot/1e6:.2f} | "
                 f"{(sum(a)/1e6 if a else float('nan')):.0f} | {(tot/sum(a) if a else float('nan')):.4f} |")
json.dump(out, open(sys.argv[3], 'w'), indent=1)
open(sys.argv[4], 'w').write('\n'.join(lines) + '\n')
print('\n'.join(lines))
