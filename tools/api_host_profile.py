#!/usr/bin/env python3
"""Whose host time is it?    tests/api_program.py -- `mpc.output(a * b)` on SecFld(2^61-1) arrays through the
unmodified mpyc runtime under mpyc_amd.install(), ONE party -- under cProfile at n = 10^4, 10^5, 10^6 (and 10^7), the timed
repetitions only.  Every function's own time (tottime) is attributed to
    mirror      code under mpyc_amd/ (HostView, frame lookup, FieldArray, ShareMatrix, engine, ctypes marshalling), including the
                time of built-ins (ctypes foreign calls, torch, numpy) called FROM there
    reference   code under mpyc/ (the runtime's coroutine machinery: mpc_coro, returnType, gather_shares, output, _reshare, ...)
                and what it calls in asyncio / the standard library
    other       everything else (the program itself, import-time leftovers)
-> one markdown table: wall ms per repetition, ms and share per group, GPU-busy ms beside it.

    python tools/api_host_profile.py [out.md]          (GPU box; needs the staged reference copy _refstage/)
"""
import json
import os
import pstats
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = next((r for r in (os.path.join(ROOT, '_refstage'), '/root/reference') if os.path.isdir(os.path.join(r, 'mpyc'))), None)
PROG = os.path.join(ROOT, 'tests', 'api_program.py')


def group_of(filename):
    f = filename.replace('\\', '/')
    if '/mpyc_amd/' in f:
        return 'mirror'
    if f.endswith('/mpyc/runtime.py') or '/mpyc/' in f:
        return 'reference'
    if '/asyncio/' in f or '/selectors.py' in f or '/concurrent/' in f:
        return 'reference'                      # the event loop runs on behalf of the runtime's coroutines
    return None


def profile(n, reps, tmp, parties=1):
    prof = os.path.join(tmp, f'api_{n}_{parties}.prof')
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([os.path.join(ROOT, 'tests'), ROOT, REF]), API_MODE=os.environ.get('HOSTPROF_MODE', 'gpu'), API_N=str(n),
               API_REPS=str(reps), API_WARMUP='3', API_CPROFILE=prof, MPYC_AMD_IPC_WIRE='1' if parties > 1 else '0')
    for k in ('API_SEED', 'API_DIGEST', 'RANK', 'WORLD_SIZE', 'LOCAL_RANK'):
        env.pop(k, None)
    r = subprocess.run([sys.executable, PROG, '--no-log'] + ([f'-M{parties}'] if parties > 1 else []), capture_output=True, text=True,
                       cwd=tmp, env=env, timeout=900)
    line = next((ln for ln in r.stdout.splitlines() if ln.startswith('API_RESULT ')), None)
    if r.returncode != 0 or line is None:
        raise SystemExit((r.stdout + r.stderr)[-2000:])
    res = json.loads(line[len('API_RESULT '):])
    st = pstats.Stats(prof)
    own = {'mirror': 0.0, 'reference': 0.0, 'other': 0.0}
    top = {'mirror': [], 'reference': []}
    for func, (cc, nc, tt, ct, callers) in st.stats.items():
        g = group_of(func[0])
        if g is not None:
            own[g] += tt
            top[g].append((tt, f'{os.path.basename(func[0])}:{func[1]} {func[2]}'))
            continue
        # built-ins and library code: charge their own time to whoever called them
        total_calls = sum(c[0] for c in callers.values()) or 1
        for caller, c in callers.items():
            share = tt * c[0] / total_calls
            cg = group_of(caller[0])
            own[cg if cg is not None else 'other'] += share
            if cg is not None and share > 0:
                top[cg].append((share, f'  {func[2]} <- {os.path.basename(caller[0])}:{caller[1]}'))
        if not callers:
            own['other'] += tt
    # (profiled repetitions include the 3 warm-up ones: API_WARMUP + API_REPS)
    nrep = reps + 3
    wall = sum(res['times_s']) / len(res['times_s'])
    return {'n': n, 'parties': parties, 'reps': reps, 'wall_ms_per_rep': wall * 1e3, 'gpu_busy_ms_per_rep': (res['gpu_busy_ms'] or 0.0) / reps,
            'own_ms_per_rep': {g: v / nrep * 1e3 for g, v in own.items()},
            'top': {g: [f'{t / nrep * 1e6:.0f} us  {name}' for t, name in sorted(v, reverse=True)[:6]] for g, v in top.items()}}


def main():
    if REF is None:
        raise SystemExit('no importable mpyc checkout (stage one with tools/stage_reference.sh)')
    out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, 'gpurun_out', 'r05_api_host.md')
    rows = []
    with tempfile.TemporaryDirectory() as tmp:
        sizes = [int(v) for v in os.environ.get('HOSTPROF_SIZES', '10000,100000,1000000,10000000').split(',')]
        for n, reps in [(n_, 200 if n_ <= 10**5 else 100 if n_ <= 10**6 else 30) for n_ in sizes]:
            rows.append(profile(n, reps, tmp))
        if os.environ.get('HOSTPROF_M3', '1') == '1' and os.environ.get('HOSTPROF_MODE', 'gpu') == 'gpu':
            for n in (10**6, 10**7):              # three local parties over the device-side wire (party 0 profiled)
                rows.append(profile(n, 30, tmp, parties=3))
    lines = ['| parties | n | wall per repetition (profiled run) | mirror (`mpyc_amd/`) | reference runtime (`mpyc/`, asyncio) | other | GPU busy |',
             '|---|---|---|---|---|---|---|']
    for r in rows:
        o = r['own_ms_per_rep']
        tot = sum(o.values()) or 1.0
        lines.append(f"| {r['parties']} | {r['n']:.0e} | {r['wall_ms_per_rep']:.3f} ms | {o['mirror']:.3f} ms ({o['mirror'] / tot:.0%}) | "
                     f"{o['reference']:.3f} ms ({o['reference'] / tot:.0%}) | {o['other']:.3f} ms | {r['gpu_busy_ms_per_rep']:.3f} ms |")
    lines.append('')
    for r in rows:
        lines.append(f"m = {r['parties']}, n = {r['n']:.0e}: largest own-time entries per repetition")
        for g in ('mirror', 'reference'):
            lines.append(f'  {g}:')
            lines += [f'    {t}' for t in r['top'][g]]
    text = '\n'.join(lines) + '\n'
    os.makedirs(os.path.dirname(out), exist_ok=True)
    with open(out, 'w') as fh:
        fh.write(text)
    print(text)


main()
