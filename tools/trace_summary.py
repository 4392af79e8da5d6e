#!/usr/bin/env python3
"""Group a rocprofv3 kernel trace by (kernel, grid): median / min duration per group, in launch order.
usage: trace_summary.py <kernel_trace.csv> [substring filter ...]"""
import csv, re, statistics, sys
rows = list(csv.DictReader(open(sys.argv[1])))
filt = sys.argv[2:] or ['ffgpu::', 'k_gf8', 'k_sbox']
groups, order = {}, []
for r in sorted(rows, key=lambda r: int(r['Start_Timestamp'])):
    nm = r['Kernel_Name']
    if not any(s in nm for s in filt):
        continue
    key = (re.sub(r'\(.*', '', nm).replace('void ffgpu::', '').replace('void ', '')[:80], r['Grid_Size_X'], r['Grid_Size_Y'])
    if key not in groups:
        groups[key] = []
        order.append(key)
    groups[key].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for key in order:
    d = groups[key]
    print(f"{statistics.median(d):9.2f} us med  {min(d):9.2f} min  x{len(d):<4} grid {key[1]:>9},{key[2]}  scratch? {key[0]}")
