#!/usr/bin/env python3
"""Regenerate profiles/r01_* from gpurun_out/{prof_TAG,pmc_TAG_*} and gpurun_out/bench.log.  usage: make_profiles.py TAG"""
import csv, os, re, shutil, subprocess, sys
tag = sys.argv[1]
G = 'gpurun_out'
subprocess.run([sys.executable, 'tools/pmc_summary.py', f'{G}/pmc_{tag}_FETCH_SIZE/FETCH_SIZE_counter_collection.csv',
                f'{G}/pmc_{tag}_WRITE_SIZE/WRITE_SIZE_counter_collection.csv', 'profiles/r01_pmc_traffic.json', '/tmp/pmc.md'],
               check=True, stdout=subprocess.DEVNULL)
with open('profiles/r01_pmc_traffic.md', 'w') as fh:
    fh.write('# Round 1 -- HBM traffic per launch from PMC counters (MI355X, n = 10^7 64-bit elements)\n\n')
    fh.write('Command (two separate passes, kernel-trace only): `rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -- '
             'python tools/pmc_probe.py` and the same with `--pmc WRITE_SIZE`.\n')
    fh.write('Corrections per MI355X_MICROARCH.md (HBM): counters are KiB; gfx950 FETCH_SIZE counts half of a 16 B/lane stream, so '
             'reads = 2 x FETCH_SIZE. `k_copy16` (exactly 80 MB in, 80 MB out) confirms both corrections (first row).\n\n')
    fh.write(open('/tmp/pmc.md').read())
    fh.write('\nTraffic equals the algorithmic byte count for every kernel (ratio 1.000): each input element is fetched once and '
             'each output written once; there are no re-reads to remove. The fused `k_split<..., 1, true, ...>` (local product + '
             'share generation) reads a, b and one coefficient row (240 MB) and writes three share rows (240 MB): the product c '
             'never reaches HBM.\n')
rows = list(csv.DictReader(open(f'{G}/prof_{tag}/{tag}_kernel_stats.csv')))
with open('profiles/r01_kernel_stats.md', 'w') as fh:
    fh.write('# Round 1 -- `rocprofv3 --kernel-trace --stats` of `python bench.py --steps 20 --warmup 3 --no-cpu-baseline` (MI355X)\n\n')
    fh.write('Raw file: profiles/r01_kernel_stats.csv (rocprofv3 `*_kernel_stats.csv`). Library kernels only below; the rest are '
             'torch RNG/fill kernels that create the synthetic inputs outside the timed region.\n')
    fh.write('Template arguments of k_split: <field policy, T, fused local product, non-temporal, lazy single reduction, in-kernel CSPRNG, factors given as recombinations (chain gate)>.\n\n')
    fh.write('| kernel | calls | avg us | min us | max us | total ms |\n|---|---|---|---|---|---|\n')
    for r in rows:
        if any(s in r['Name'] for s in ('ffgpu::', 'k_copy16', 'k_sbox', 'k_gf')):
            nm = re.sub(r'\(.*', '', r['Name']).replace('void ffgpu::', '').replace('void ', '')
            fh.write(f"| `{nm}` | {r['Calls']} | {float(r['AverageNs'])/1e3:.2f} | {float(r['MinNs'])/1e3:.2f} | "
                     f"{float(r['MaxNs'])/1e3:.2f} | {float(r['TotalDurationNs'])/1e6:.3f} |\n")
    fh.write('\n## bench.py line of the same build (separate, unprofiled run)\n\n```\n')
    for line in open(f'{G}/bench.log'):
        if line.startswith('{'):
            fh.write(line)
    fh.write('```\n')
shutil.copy(f'{G}/prof_{tag}/{tag}_kernel_stats.csv', 'profiles/r01_kernel_stats.csv')
print('profiles regenerated from', tag)
