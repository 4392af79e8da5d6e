#!/usr/bin/env python3
import json, sys
for f in sys.argv[1:]:
    for line in open(f):
        if line.startswith('{'):
            d = json.loads(line)
            print(f, 'value=%.2fG %s ms/step=%.4f n_gpus=%d' % (d['value'] / 1e9, d['unit'], d['ms_per_step'], d['n_gpus']))
            if 'roofline' in d:
                print('  roofline:', d['roofline'])
            for k, v in d.get('kernels', {}).items():
                print(f"  {k:28s} {v['ms_per_launch']*1e3:9.1f} us {v['achieved']:8.1f} GB/s frac={v['frac']:.3f} units/s={v.get('units_per_s', v.get('gates_per_s'))}")
            if 'cpu_baseline' in d:
                print('  cpu:', d['cpu_baseline'])
