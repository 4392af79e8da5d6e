#!/usr/bin/env python3
"""bench_detail.json (written by bench.py beside the compact line) -> one markdown table of every measured row.
usage: bench_table.py gpurun_out/bench_detail.json profiles/r04_bench_table.md [title]"""
import json, sys

d = json.load(open(sys.argv[1]))
title = sys.argv[3] if len(sys.argv) > 3 else 'bench.py rows'
L = [f'# {title}', '',
     f"Headline: **{d['value']:.4g} {d['unit']}**, {d['ms_per_step']} ms/step, {d['n_gpus']} GPU, steps {d['steps']}, warm-up {d['warmup']}; "
     f"workload: {d['config']['workload']}.", '']
rf = d.get('roofline', {})
if rf:
    L += [f"Dominant kernel `{rf.get('kernel')}`: {rf.get('ms_per_launch')} ms per launch, {rf.get('achieved')} {rf.get('unit')} = "
          f"**{rf.get('frac')}** of {rf.get('peak')} ({rf.get('frac_of_measured_copy')} x the device copy of the same run); algorithmic "
          f"{rf.get('bytes_per_launch')} B, PMC traffic {rf.get('traffic')} B.", '']
cb = d.get('cpu_baseline', {})
if cb:
    L += [f"CPU baseline ({cb.get('kind')}): {cb.get('value'):.4g} {cb.get('unit')} on {cb.get('procs', cb.get('cores'))} processes of "
          f"{cb.get('host_cores')} host cores; 1 core {cb.get('value_1core'):.4g}; C port {cb.get('port_value', 0):.4g} on "
          f"{cb.get('port_cores')} threads.", '']
L += ['| row | ms/launch | achieved | frac of HBM (or of int8 MFMA) | bound | valu_frac | units/s | bytes/unit |', '|---|---|---|---|---|---|---|---|']
for k, r in d.get('kernels', {}).items():
    if not isinstance(r, dict):
        continue
    L.append(f"| `{k}` | {r.get('ms_per_launch', '')} | {r.get('achieved', '')} {r.get('unit', '')} | {r.get('frac', '')} | {r.get('bound', '')} | "
             f"{r.get('valu_frac', '')} | {r.get('units_per_s', ''):.4g} | {r.get('algorithmic_bytes_per_unit', '')} |"
             if isinstance(r.get('units_per_s'), (int, float)) else
             f"| `{k}` | {r.get('ms_per_launch', '')} | {r.get('achieved', '')} {r.get('unit', '')} | {r.get('frac', '')} | {r.get('bound', '')} | "
             f"{r.get('valu_frac', '')} |  | {r.get('algorithmic_bytes_per_unit', '')} |")
for sec in ('configs2', 'api', 'multi_gpu'):
    if isinstance(d.get(sec), dict):
        L += ['', f'## `{sec}`', '', '```json', json.dumps(d[sec], indent=1)[:6000], '```']
open(sys.argv[2], 'w').write('\n'.join(L) + '\n')
print(f'{len(d.get("kernels", {}))} rows -> {sys.argv[2]}')
