#!/usr/bin/env python3
"""rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES counter CSV of tools/valu_probe.py -> VALU instructions per element for every row.
usage: valu_summary.py counter_collection.csv valu_order.json out.json out.md
The probe opens each row with a k_copy16 launch (marker); the library dispatches between two markers belong to that row
(a row is one kernel, or a kernel plus its generator-state update).  SQ_INSTS_VALU counts wave instructions: per element =
64 x instructions / ... no: lane-operations per element = SQ_INSTS_VALU x 64 / n; issue slots per element = SQ_INSTS_VALU x 64 / n
as well (one wave instruction occupies its SIMD for 4 cycles x 16 lanes)."""
import collections, csv, json, re, sys

rows = json.load(open(sys.argv[2]))
n_default, per = rows['n'], rows['launches_per_row']
units = rows.get('units', {})
disp = collections.OrderedDict()          # dispatch id -> {name, counters}
for r in csv.DictReader(open(sys.argv[1])):
    name = re.sub(r'\(.*', '', r['Kernel_Name']).replace('void ffgpu::', '').replace('void ', '')
    d = disp.setdefault(int(r['Dispatch_Id']), {'name': name, 'grid': int(r['Grid_Size'])})
    d[r['Counter_Name']] = d.get(r['Counter_Name'], 0.0) + float(r['Counter_Value'])
seq = [d for _, d in sorted(disp.items()) if d['name'].startswith('k_')]
# split at the markers; the probe's own sequence is the LAST len(rows) groups (set-up may launch library kernels before)
groups, cur = [], None
for d in seq:
    if d['name'].startswith('k_copy16'):
        cur = []
        groups.append(cur)
    elif cur is not None:
        cur.append(d)
groups = groups[-len(rows['rows']):]
out, lines = {}, ['| bench row | kernel | VALU wave-instructions per launch | per element (x 64 lanes / n) | waves |', '|---|---|---|---|---|']
for name, g in zip(rows['rows'], groups):
    main = [d for d in g if d.get('SQ_WAVES', 0) > 64]          # (the 1-wave generator-state update is not the row's kernel)
    if not main:
        continue
    insts = sum(d.get('SQ_INSTS_VALU', 0.0) for d in main) / per
    kern = main[0]['name']
    n = units.get(name, n_default)
    out[name] = {'kernel': kern, 'sq_insts_valu_per_launch': round(insts, 1), 'valu_lane_ops_per_unit': round(insts * 64 / n, 2),
                 'sq_waves': main[0].get('SQ_WAVES'), 'n': n}
    lines.append(f"| `{name}` | `{kern}` | {insts:.4g} | {insts * 64 / n:.1f} | {main[0].get('SQ_WAVES', 0):.0f} |")
import os
out['_meta'] = {'commit': os.environ.get('VALU_COMMIT'), 'counter': 'SQ_INSTS_VALU (wave instructions) x 64 / units'}
json.dump(out, open(sys.argv[3], 'w'), indent=1)
open(sys.argv[4], 'w').write('\n'.join(lines) + '\n')
print('\n'.join(lines))
