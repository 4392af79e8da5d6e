#!/usr/bin/env python3
"""Regenerate profiles/<TAG>_* from gpurun_out/ (written there by tools/gpu_r02.sh / tools/gpu_prof.sh).

    usage: make_profiles.py TAG [stats] [pmc] [aes]

  stats  profiles/TAG_kernel_stats.{md,csv}   from gpurun_out/prof_TAG/TAG_kernel_stats.csv + gpurun_out/bench.log
  pmc    profiles/TAG_pmc_traffic.{md,json}   from gpurun_out/pmc_TAG_{FETCH,WRITE}_SIZE/
  aes    profiles/TAG_np_aes_dropin.md        from gpurun_out/prof_TAG_aes_m{1,3}/ (the reference's np_aes.py run
                                              unmodified under mpyc_amd.install(), one rocprofv3 table per party process)
"""
import csv
import glob
import os
import re
import shutil
import subprocess
import sys

tag = sys.argv[1]
what = set(sys.argv[2:]) or {'stats', 'pmc', 'aes'}
G = 'gpurun_out'
OURS = ('ffgpu::', 'k_copy16', 'k_sbox', 'k_gf', 'k_prss', 'k_limb', 'k_mat', 'k_vec', 'k_splitk')


def short(name):
    nm = re.sub(r'\(.*', '', name).replace('void ffgpu::', '').replace('void ', '')
    return nm if len(nm) < 150 else nm[:147] + '...'


def table(fh, rows, only_ours=True, top=None):
    fh.write('| kernel | calls | avg us | min us | max us | total ms |\n|---|---|---|---|---|---|\n')
    k = 0
    for r in rows:
        if only_ours and not any(s in r['Name'] for s in OURS):
            continue
        fh.write(f"| `{short(r['Name'])}` | {r['Calls']} | {float(r['AverageNs'])/1e3:.2f} | {float(r['MinNs'])/1e3:.2f} | "
                 f"{float(r['MaxNs'])/1e3:.2f} | {float(r['TotalDurationNs'])/1e6:.3f} |\n")
        k += 1
        if top and k >= top:
            break


if 'pmc' in what and os.path.isdir(f'{G}/pmc_{tag}_FETCH_SIZE'):
    subprocess.run([sys.executable, 'tools/pmc_summary.py', f'{G}/pmc_{tag}_FETCH_SIZE/FETCH_SIZE_counter_collection.csv',
                    f'{G}/pmc_{tag}_WRITE_SIZE/WRITE_SIZE_counter_collection.csv', f'profiles/{tag}_pmc_traffic.json', '/tmp/pmc.md'],
                   check=True, stdout=subprocess.DEVNULL)
    with open(f'profiles/{tag}_pmc_traffic.md', 'w') as fh:
        fh.write(f'# {tag} -- HBM traffic per launch from PMC counters (MI355X, n = 10^7 elements)\n\n')
        fh.write('Command (two separate passes, kernel-trace only): `rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -- '
                 'python tools/pmc_probe.py` and the same with `--pmc WRITE_SIZE`.\n')
        fh.write('Corrections per MI355X_MICROARCH.md (HBM): counters are KiB; gfx950 FETCH_SIZE counts half of a 16 B/lane stream, so '
                 'reads = 2 x FETCH_SIZE. `k_copy16` (exactly 80 MB in, 80 MB out) confirms both corrections (first row).\n\n')
        fh.write(open('/tmp/pmc.md').read())

if 'stats' in what and os.path.exists(f'{G}/prof_{tag}/{tag}_kernel_stats.csv'):
    rows = list(csv.DictReader(open(f'{G}/prof_{tag}/{tag}_kernel_stats.csv')))
    with open(f'profiles/{tag}_kernel_stats.md', 'w') as fh:
        fh.write(f'# {tag} -- `rocprofv3 --kernel-trace --stats` of `python bench.py --steps 20 --warmup 3 --no-cpu-baseline` (MI355X)\n\n')
        fh.write(f'Raw file: profiles/{tag}_kernel_stats.csv (rocprofv3 `*_kernel_stats.csv`). Library kernels only below; the rest are '
                 'torch RNG/fill kernels that create the synthetic inputs outside the timed region.\n')
        fh.write('Template arguments of k_split: <field policy, T, fused local product, non-temporal, '
                 'in-kernel CSPRNG, factors given as recombinations (chain gate)>.\n\n')
        table(fh, rows)
        fh.write('\n## bench.py line of the same build (separate, unprofiled run)\n\n```\n')
        for line in open(f'{G}/bench.log'):
            if line.startswith('{'):
                fh.write(line)
        fh.write('```\n')
    shutil.copy(f'{G}/prof_{tag}/{tag}_kernel_stats.csv', f'profiles/{tag}_kernel_stats.csv')

if 'aes' in what and glob.glob(f'{G}/prof_{tag}_aes_m1/*/*kernel_stats.csv'):
    with open(f'profiles/{tag}_np_aes_dropin.md', 'w') as fh:
        fh.write(f'# {tag} -- the reference\'s own `demos/np_aes.py`, unmodified, under `mpyc_amd.install()` on an MI355X\n\n')
        fh.write('Command (tools/gpu_r02.sh; `_refstage/` is an untracked copy of the mpyc checkout that travels to the GPU box):\n\n```\n'
                 'export PYTHONPATH=$R/mpyc_amd/autoinstall:$R:$R/_refstage MPYC_GPU=1 MPYC_AMD_TRACE_INSTALL=1\n'
                 'cd _refstage/demos && rocprofv3 --kernel-trace --stats --output-format csv -- python np_aes.py -1 [-M3]\n```\n\n'
                 '`MPYC_GPU=1` makes `mpyc_amd/autoinstall/sitecustomize.py` call `mpyc_amd.install()` in every party process '
                 '(mpyc spawns parties 1..m-1 itself, runtime.py:5157-5189; rocprofv3 follows them, one table per pid below). '
                 'Ciphertext expected: `69c4e0d86a7b0430d8cdb78070b4c55a` (FIPS-197 C.1, docs/demos.rst:611).\n')
        for mm in ('m1', 'm3'):
            log = f'{G}/rocprof_{tag}_aes_{mm}.log'
            fh.write(f'\n## np_aes.py -1 {"-M3" if mm == "m3" else ""}  -- program output\n\n```\n')
            for line in open(log):
                if not re.match(r'^[EWI]\d{8} ', line) and line.strip():
                    fh.write(line)
            fh.write('```\n')
            for f in sorted(glob.glob(f'{G}/prof_{tag}_aes_{mm}/*/*kernel_stats.csv')):
                rows = list(csv.DictReader(open(f)))
                pid = os.path.basename(f).split('_')[0]
                ours = [r for r in rows if any(s in r['Name'] for s in OURS)]
                tot = sum(float(r['TotalDurationNs']) for r in rows) / 1e6
                tot_ours = sum(float(r['TotalDurationNs']) for r in ours) / 1e6
                fh.write(f'\n### party process pid {pid}: {sum(int(r["Calls"]) for r in ours)} launches of libffgpu kernels '
                         f'({tot_ours:.2f} ms of {tot:.2f} ms GPU time; the rest: hipMemcpy staging `__amd_rocclr_copyBuffer` '
                         'and torch data-movement kernels)\n\n')
                table(fh, rows, only_ours=True, top=24)
        pt = f'{G}/pytest_gpu.log'
        if os.path.exists(pt):
            fh.write('\n## `python -m pytest tests -m gpu -x -q` of the same call (includes tests/test_mpyc_dropin.py: the '
                     'reference\'s whole test suite + np_aes -1 / -1 -M3 under install() on the kernels)\n\n```\n')
            fh.write(''.join(open(pt).readlines()[-8:]))
            fh.write('```\n')
print('profiles regenerated from', tag)
