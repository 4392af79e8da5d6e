#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd (sqlite) kernel trace into the familiar --stats table.
usage: rocpd_stats.py results.db [out.md]"""
import re, sqlite3, sys
db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
namecol = 'name' if 'name' in cols else [c for c in cols if 'name' in c][0]
rows = cur.execute(f"select {namecol}, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                   f"from kernels group by {namecol} order by 3 desc").fetchall()
total = sum(r[2] for r in rows) or 1
lines = ['| kernel | calls | total ms | avg us | min us | max us | % |', '|---|---|---|---|---|---|---|']
for name, calls, tot, avg, mn, mx in rows:
    short = re.sub(r'\(.*', '', name)
    short = short.replace('void ffgpu::', '').replace('void ', '')
    lines.append(f'| `{short[:110]}` | {calls} | {tot/1e6:.3f} | {avg/1e3:.2f} | {mn/1e3:.2f} | {mx/1e3:.2f} | {100*tot/total:.1f} |')
out = '\n'.join(lines)
print(out)
if len(sys.argv) > 2:
    open(sys.argv[2], 'w').write(out + '\n')
