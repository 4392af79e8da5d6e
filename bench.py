#!/usr/bin/env python3
"""bench.py -- field-ops/sec of the MPyC secure-multiplication hot path on MI355X.

Workload (BASELINE.json configs[1], per GPU): a SecFld(GF(2^61-1)) array of n = 10^7
elements.  One STEP is one pass of the path the reference runs for `a * b` on secure arrays
(runtime.py:1096-1141 np_multiply -> :603-689 _reshare), for the default 3-party setting
m=3, t=1:
    c      = a * b                      n mod-muls        (finfields.py:1105-1112)
    shares = np_random_split(c, t, m)   n share gens      (thresha.py:47-64, coefficients supplied)
    y      = np_recombine(2t+1 rows)    n recombinations  (thresha.py:119-132)
=> 3n field-ops per step (one op = one element through one stage).  The first two stages run as
ONE kernel (ffgpu_mul_split): the product is formed in registers and goes straight into share
generation, exactly what mpyc_amd.finfields does for `a * b` followed by np_random_split (deferred
product); c itself is never written to HBM.  The unfused three-kernel step is timed as well and
reported as `unfused` (and each kernel separately under `kernels`).  Inputs are resident in
HBM before the timed region; several independent buffer sets are rotated so that no launch
finds its operands in the 256 MiB Infinity Cache.

Contract: `python bench.py --gpus N --steps K --warmup W` (torchrun for N>1, one rank per
GPU); W untimed warm-up steps, exactly K timed steps between barrier+synchronize, MAX over
ranks, rank 0 prints ONE JSON line.  Elements shard across ranks with no collective on the
data path (every element is independent): weak scaling, n per GPU fixed.

Extra objects on the line:
  roofline      dominant kernel: algorithmic bytes per launch / mean launch time measured with
                events on the launch stream inside this run; peak = 8 TB/s HBM (spec).
  kernels       the same for every kernel of the step, plus configs[2] (P64, m=7, t=3) and the
                measured device-copy bandwidth (achievable-HBM yardstick).
  cpu_baseline  oracle/fforacle.c (C port of the reference path) timed on this host.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np
import torch

P61 = 2**61 - 1
P64 = 2**64 - 189
HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6.3 TB/s measured copy
T_PROCESS_START = time.perf_counter()
SECTIONS_S = {}            # wall seconds per section of this run (`sections_s` in the detail; `wall_s` on the line)


class section:
    """with section('name'): ... -- wall time of a section of the run, accumulated into SECTIONS_S."""

    def __init__(self, name):
        self.name = name

    def __enter__(self):
        self.t0 = time.perf_counter()
        return self

    def __exit__(self, *exc):
        SECTIONS_S[self.name] = round(SECTIONS_S.get(self.name, 0.0) + time.perf_counter() - self.t0, 2)
        return False


_LAP = [T_PROCESS_START]


def lap(name):
    """Wall time since the previous lap() (or process start) goes to section `name`."""
    now = time.perf_counter()
    SECTIONS_S[name] = round(SECTIONS_S.get(name, 0.0) + now - _LAP[0], 2)
    _LAP[0] = now


def uniform_field(gen, n, p, device):
    """n uniform canonical elements of GF(p), p < 2^64, generated on the device."""
    nb = p.bit_length()
    x = torch.randint(-2**63, 2**63 - 1, (n,), dtype=torch.int64, device=device, generator=gen)
    if nb < 64:
        x = (x >> (64 - nb)) & ((1 << nb) - 1)
        x = torch.where(x >= p, x - p, x)
    else:
        # unsigned compare x >= p on int64 bit patterns: p = 2^64 - c  <=>  signed x in [-c, -1]
        c = (1 << 64) - p
        x = torch.where((x < 0) & (x >= -c), x + c, x)
    return x


class StepData:
    """One rotating buffer set for the gate pass."""

    def __init__(self, ctx, n, t, m, gen):
        from mpyc_amd.engine import DevArray
        dev = ctx.torch_device
        p = ctx.modulus
        self.a = DevArray(ctx, uniform_field(gen, n, p, dev), n)
        self.b = DevArray(ctx, uniform_field(gen, n, p, dev), n)
        self.c = ctx.empty(n)
        self.coef = ctx.empty_matrix(max(t, 1), n)
        for j in range(t):
            self.coef.row(j).t.copy_(uniform_field(gen, n, p, dev))
        self.shares = ctx.empty_matrix(m, n)
        self.y = ctx.empty(n)
        self.rec = None     # pre-marshalled recombination launch


def time_launches(fn, sets, reps):
    """Mean ms per launch of fn(set) over reps passes through all sets (events on the launch stream)."""
    for s in sets:
        fn(s)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        for s in sets:
            fn(s)
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / (reps * len(sets))


def roof(bytes_per_launch, ms):
    gbs = bytes_per_launch / (ms * 1e-3) / 1e9
    return {'bound': 'hbm', 'achieved': round(gbs, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
            'frac': round(gbs / HBM_PEAK_GBS, 4), 'ms_per_launch': round(ms, 5),
            'bytes_per_launch': int(bytes_per_launch), 'traffic': None}


I8_MFMA_PEAK_TOPS = 3944.0   # measured dense int8 MFMA ceiling (MI355X_MICROARCH.md, matrix-core table; 2 ops per MAC)


def roof_mfma(field_macs, digits, ms):
    """Matrix-core product: every field MAC is digits x digits int8 MACs on the MFMA pipe (signed base-256 digit
    planes, all 2L-1 diagonals); achieved = int8 ops actually issued / time, against the int8 MFMA ceiling."""
    tops = 2.0 * field_macs * digits * digits / (ms * 1e-3) / 1e12
    return {'bound': 'mfma', 'achieved': round(tops, 1), 'peak': I8_MFMA_PEAK_TOPS, 'unit': 'TOP/s (int8)',
            'frac': round(tops / I8_MFMA_PEAK_TOPS, 4), 'ms_per_launch': round(ms, 4), 'traffic': None,
            'int8_macs_per_field_mac': digits * digits, 'units_per_s': round(field_macs / (ms * 1e-3), 1),
            'field_GMAC_per_s': round(field_macs / (ms * 1e-3) / 1e9, 1)}


COMPACT_LIMIT = 8000      # the driver's stdout tail holds ~8 KB: the FINAL line must fit it whole


# ---- bounded sub-processes: the API-level legs and the reference baseline run OTHER programs (party processes of the
# reference runtime, worker pools).  Each gets its own process group -- a timeout ends the parties it spawned as well, so
# nothing lingers on the GPU or on a TCP port of the next leg -- a tight limit of its own, and they all share one budget:
# when it is spent the remaining legs are reported as skipped and the line goes out (the driver's round-5 run took 977 s
# because ONE leg waited for its 900 s limit).
LEG_BUDGET = {'deadline': None, 'leg_timeout': 60.0}


def budget_left():
    return float('inf') if LEG_BUDGET['deadline'] is None else LEG_BUDGET['deadline'] - time.perf_counter()


def run_bounded(cmd, env, timeout, cwd='/tmp'):
    """-> (returncode, stdout, stderr), or None when the program did not finish within min(timeout, budget left)."""
    import signal
    import subprocess
    limit = min(timeout, budget_left())
    if limit < 3:
        return None
    p = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, cwd=cwd, env=env, start_new_session=True)
    try:
        out, err = p.communicate(timeout=limit)
        return p.returncode, out, err
    except subprocess.TimeoutExpired:
        return None
    finally:
        try:
            os.killpg(p.pid, signal.SIGKILL)          # the whole group: party processes spawned by party 0 (runtime.py:5171-5189)
        except (ProcessLookupError, PermissionError):
            pass
        try:
            p.communicate(timeout=5)
        except Exception:          # noqa: BLE001
            pass


_PORT_NEXT = [0]


def free_base_port(count=4):
    """A base port for one multi-party leg (`-B`): `count` consecutive ports that are free right now, a fresh range per leg
    (the runtime's default 11365.. would be shared by consecutive legs: a party of the previous leg that is still shutting
    down makes the next leg's connection loop, runtime.py:263-285, spin until the limit)."""
    import socket
    for _ in range(200):
        base = 20000 + (os.getpid() * 7 + _PORT_NEXT[0] * 16) % 20000
        _PORT_NEXT[0] += 1
        ok = True
        for q in range(count):
            with socket.socket() as sk:
                try:
                    sk.bind(('127.0.0.1', base + q))
                except OSError:
                    ok = False
                    break
        if ok:
            return base
    return 11365


def _pick(d, keys):
    return {k_: d[k_] for k_ in keys if isinstance(d, dict) and k_ in d}


def compact_line(out):
    """The ONE JSON line the driver parses (last line of stdout): the contract keys, `roofline`, `cpu_baseline` and short
    summaries of the API / multi-GPU sections.  Every per-kernel row, the full `api`, `configs2` and `multi_gpu` objects
    travel in bench_detail.json (and in an earlier, `# detail `-prefixed stdout line), never here: the final line stays
    far below COMPACT_LIMIT characters whatever the run measured (tests/test_bench_line.py)."""
    line = _pick(out, ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling',
                       'vs_baseline', 'dtype', 'data'))
    cfg = out.get('config', {})
    line['config'] = _pick(cfg, ('workload', 'n_per_gpu', 'prime', 'm', 't', 'k', 'field_ops_per_step', 'buffer_sets',
                                 'parallelism'))
    if len(line['config'].get('workload', '')) > 400:
        line['config']['workload'] = line['config']['workload'][:400]
    if 'roofline' in out:
        line['roofline'] = _pick(out['roofline'], ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic', 'traffic_source', 'kernel',
                                                   'name', 'ms_per_launch', 'bytes_per_launch', 'frac_of_measured_copy'))
        if 'traffic_source' in line['roofline']:
            line['roofline']['traffic_source'] = str(line['roofline']['traffic_source'])[:120]
    cb = out.get('cpu_baseline')
    if isinstance(cb, dict):
        line['cpu_baseline'] = _pick(cb, ('value', 'unit', 'cores', 'kind', 'procs', 'host_cores', 'value_1core', 'port_value',
                                          'port_cores', 'error'))
        if 'sample' in cb:
            line['cpu_baseline']['sample'] = cb['sample'][:360]
    if 'unfused' in out:
        line['unfused'] = _pick(out['unfused'], ('value', 'ms_per_step'))
    if 'mulmod_per_s_1gpu' in out:
        line['mulmod_per_s_1gpu'] = out['mulmod_per_s_1gpu']
    c2 = out.get('configs2')
    if isinstance(c2, dict):
        line['configs2'] = {'workload': 'configs[2]: GF(2^64-189), 10^7 secrets, split m=7,t=3 + recombine k',
                            'k4_secrets_per_s': c2.get('k4', {}).get('value'),
                            'k7_secrets_per_s': c2.get('k7', {}).get('value'),
                            'roofline': _pick(c2.get('roofline', {}), ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic',
                                                                        'name', 'ms_per_launch'))}
    api = out.get('api')
    if isinstance(api, dict):
        line['api'] = _pick(api, ('elements_per_s', 'gpu_busy_frac', 'gpu_busy_frac_1e8', 'vs_reference_m1',
                                  'gpu_busy_frac_m3_1e7_ipc', 'error'))
        if isinstance(api.get('m3_1e7_ipc'), dict):
            line['api']['m3_1e7_ipc'] = _pick(api['m3_1e7_ipc'], ('ms_per_rep', 'elements_per_s'))
        if isinstance(api.get('m1_1e7'), dict):
            line['api']['m1_1e7'] = _pick(api['m1_1e7'], ('ms_per_rep', 'elements_per_s'))
        for leg in ('fxp_m1_1e6', 'fxp_m1_1e6_chacha'):
            if isinstance(api.get(leg), dict):
                line['api'][leg] = _pick(api[leg], ('s_per_product', 's_per_product_and_opening', 'outliers_reference_trunc_mask', 'max_abs_error', 'error', 'skipped'))
    dd = out.get('distributed', {})
    line['distributed'] = _pick(dd, ('backend', 'world_size', 'collective_library', 'rccl_version', 'distinct_devices'))
    ranks = dd.get('ranks') or []
    line['distributed']['ranks'] = [_pick(r_, ('rank', 'local_rank', 'device', 'pci_bus_id')) for r_ in ranks[:16]]
    mg = out.get('multi_gpu')
    if isinstance(mg, dict):
        mgl = _pick(mg, ('error',))
        if isinstance(mg.get('parity'), dict):
            mgl['parity_passed_on_every_rank'] = mg['parity'].get('passed_on_every_rank')
        if isinstance(mg.get('config'), dict):
            mgl['workload'] = str(mg['config'].get('workload', ''))[:160]
            mgl['backend'] = mg['config'].get('backend')
        for leg in ('gate_sharded', 'party_major_all_to_all', 'party_major_all_to_all_pipelined', 'party_major_allgather'):
            if isinstance(mg.get(leg), dict):
                mgl[leg] = _pick(mg[leg], ('ms_per_step', 'gates_per_s', 'secrets_per_s', 'frac_of_hbm_peak', 'exchange_share',
                                           'rank0_exchange_GBps', 'exchange_GBps_per_rank', 'error'))
        line['multi_gpu'] = mgl
    for key_ in ('extras_error', 'detail_file'):
        if key_ in out:
            line[key_] = str(out[key_])[:300]
    if 'wall_s' in out:
        line['wall_s'] = out['wall_s']
    c0 = out.get('configs0')
    if isinstance(c0, dict):
        line['configs0'] = _pick(c0, ('workload', 'mirror_secrets_per_s', 'reference_secrets_per_s', 'parity', 'error', 'skipped'))
    if isinstance(out.get('valu_peak'), dict):
        line['valu_peak'] = _pick(out['valu_peak'], ('lane_ops_per_s', 'vop3_lane_ops_per_s', 'mad_u64_u32_lane_ops_per_s', 'vop3_ceiling',
                                                     'shader_clock_mhz', 'shader_clock_mhz_under_full_valu_load', 'source'))
    kern = out.get('kernels')
    if isinstance(kern, dict):
        # one number per kernel row, in TWO maps that never mix: hbm_fracs = algorithmic bytes / time / 8 TB/s for the rows
        # whose limiter is HBM, valu_fracs = VALU issue slots / time / the issue rate measured in this run for the rows whose
        # limiter is the VALU.  A row appears in exactly one; rows with no roofline meaning (host- or PCIe-bound rows,
        # launch-latency comparisons, whole-protocol timings) appear in neither -- they are in bench_detail.json.
        hbm, valu = {}, {}
        for q, row in kern.items():
            if not isinstance(row, dict):
                continue
            bound = row.get('bound')
            if bound in ('valu', 'lds+valu') and row.get('valu_frac') is not None:
                valu[q] = row['valu_frac']
            elif bound == 'hbm' and row.get('frac'):
                hbm[q] = row['frac']
        line['valu_fracs'], line['hbm_fracs'] = valu, hbm
        if len(json.dumps(line)) > COMPACT_LIMIT - 500:   # keep what fits (the VALU rows first: there are few), say what was cut
            total = len(hbm)
            line['hbm_fracs_rows_cut'] = total
            names = list(hbm)
            while names and len(json.dumps(line)) > COMPACT_LIMIT - 500:
                for q in names[-8:]:
                    hbm.pop(q)
                del names[-8:]
            line['hbm_fracs_rows_cut'] = total - len(hbm)
            if len(json.dumps(line)) > COMPACT_LIMIT - 500:
                line.pop('valu_fracs')
    text = json.dumps(line)
    for drop in ('multi_gpu', 'configs2', 'api', 'unfused'):       # never reached in practice; the cap is unconditional
        if len(text) <= COMPACT_LIMIT - 200:
            break
        line.pop(drop, None)
        text = json.dumps(line)
    return text


def emit(out, detail_path=None):
    """stdout protocol: `# detail {...}` (everything measured; not a bare JSON line), then the compact line LAST."""
    lap('rest')
    out['sections_s'] = dict(SECTIONS_S)
    out['wall_s'] = round(time.perf_counter() - T_PROCESS_START, 1)
    paths = [detail_path] if detail_path else [os.path.join(ROOT, 'bench_detail.json'),
                                                os.path.join(ROOT, 'gpurun_out', 'bench_detail.json')]
    written = []
    for path in paths:
        try:
            os.makedirs(os.path.dirname(path), exist_ok=True)
            with open(path, 'w') as fh:
                json.dump(out, fh, indent=1)
            written.append(os.path.relpath(path, ROOT))
        except OSError:
            pass
    if written:
        out['detail_file'] = written[0]
    sys.stdout.write('# detail ' + json.dumps(out) + '\n')
    sys.stdout.write(compact_line(out) + '\n')
    sys.stdout.flush()


VALU_COUNTS_FILES = ('r06_valu.json', 'r05_valu.json', 'r04_valu.json')     # SQ_INSTS_VALU per unit of every VALU-bound row (tools/valu_probe.py)


def measure_valu_peak(ctx):
    """The chip's integer-VALU issue rates, MEASURED in this run (ffgpu_valu_probe: 8 independent dependent-chains per wave, four
    waves per SIMD; profiles/r05_valu_rates.md has the whole table).  Two-operand VOP2 instructions issue at ~2.35 cycles per
    wave64 once a SIMD holds two or more waves, three-operand VOP3 ones and v_mad_u64_u32 at ~4.1-4.4.  The PEAK a row is priced
    against is the two-operand rate times the HIGHEST shader clock any probe of this run reached (a pure-VALU load at full
    occupancy clocks lowest): the rate no instruction mix can exceed, so a `valu_frac` above 1 would mean the instruction count
    is wrong; a kernel made of three-operand instructions tops out at `vop3_ceiling` of it.  No clock is assumed anywhere."""
    probes = {}
    for name, op in (('xor_b32', 3), ('add_u32', 1), ('bitop3_b32', 0), ('mad_u64_u32', 2)):
        for waves in (4, 1):
            rate, mhz, _ = ctx.valu_probe(op, waves_per_simd=waves)
            probes[f'{name}_w{waves}'] = {'lane_ops_per_s': round(rate, 1), 'shader_clock_mhz': round(mhz, 1)}
    clock_max = max(p_['shader_clock_mhz'] for p_ in probes.values())

    def at_max(key):
        return probes[key]['lane_ops_per_s'] * clock_max / probes[key]['shader_clock_mhz']
    fast = max(at_max('xor_b32_w4'), at_max('add_u32_w4'))
    return {'lane_ops_per_s': round(fast, 1), 'vop3_lane_ops_per_s': round(at_max('bitop3_b32_w4'), 1),
            'mad_u64_u32_lane_ops_per_s': round(at_max('mad_u64_u32_w4'), 1), 'vop3_ceiling': round(at_max('bitop3_b32_w4') / fast, 3),
            'shader_clock_mhz': clock_max, 'shader_clock_mhz_under_full_valu_load': probes['bitop3_b32_w4']['shader_clock_mhz'],
            'probes': probes,
            'source': 'ffgpu_valu_probe in this run: two-operand (VOP2) rate at 4 waves per SIMD x the highest shader clock seen'}


def annotate_valu(kern, out):
    """Rows whose limiter is the VALU (in-kernel ChaCha, carry-less products, exponentiations, LDS-table recombination):
    `bound: "valu"` and `valu_frac` = VALU instructions per unit (rocprofv3 --pmc SQ_INSTS_VALU of tools/valu_probe.py,
    profiles/r05_valu.json -- counted at the commit named in its `_meta`, reported as `valu_counts`) x 64 lanes x units/s / the
    issue rate MEASURED in this run (out['valu_peak']: the two-operand rate, an upper bound for any mix -- three-operand
    instructions and multiplies issue at `vop3_ceiling` of it, so a kernel made of those is issue-bound at that fraction).
    `frac` stays the fraction of the HBM peak for the row's bytes."""
    peak = out.get('valu_peak')
    path = next((os.path.join(ROOT, 'profiles', f_) for f_ in VALU_COUNTS_FILES if os.path.exists(os.path.join(ROOT, 'profiles', f_))), None)
    if not isinstance(peak, dict) or path is None:
        return
    with open(path) as fh:
        data = json.load(fh)
    meta = data.pop('_meta', {})
    out['valu_counts'] = {'file': os.path.relpath(path, ROOT), 'counted_at_commit': meta.get('commit'),
                          'note': 'instruction counts are a property of the kernels at that commit; the RATES are from this run'}
    fast = peak['lane_ops_per_s']
    jobs = [(row, info) for row, info in data.items()]
    jobs += [(row, data[r_['valu_counts_of']]) for row, r_ in kern.items()
             if isinstance(r_, dict) and r_.get('valu_counts_of') in data]          # rows that run a counted kernel another way
    for row, info in jobs:
        r = kern.get(row)
        if not isinstance(r, dict) or not r.get('ms_per_launch') or not r.get('units_per_s'):
            continue
        ops = float(info['valu_lane_ops_per_unit'])
        # (the dense GF(2^n) recombination waits on its LDS look-ups as much as on the VALU: both are named)
        r.update(bound='lds+valu' if row.endswith('_dense') else 'valu', valu_lane_ops_per_unit=ops,
                 valu_frac=round(ops * r['units_per_s'] / fast, 4))


def cpu_baseline(n_full, t, m, lam, seed=20260925):
    """The reference's own CPU path on this host (kind "reference": mpyc's FiniteFieldArray.__mul__,
    thresha.np_random_split with live secrets.randbelow draws, thresha.np_recombine -- oracle/refbaseline.py -- on 1
    core and as one process per core over equal slices), when an mpyc checkout is importable (the staged copy
    _refstage/ on the GPU box, /root/reference in the build container).  The C port of the same pass
    (oracle/fforacle.c, OpenMP) is timed beside it on the full workload and reported as `port_value` -- a much harder
    baseline than the reference; it is `value` only when no reference checkout is present (kind "port").
    Checker/baseline only -- never the product."""
    from oracle import coracle, refbaseline
    rng = np.random.default_rng(seed)
    cores = coracle.max_threads()
    n = n_full
    a = rng.integers(0, P61, size=n, dtype=np.uint64)
    b = rng.integers(0, P61, size=n, dtype=np.uint64)
    coef = rng.integers(0, P61, size=(t, n), dtype=np.uint64)
    cf = coracle.CField(P61)

    def one_pass():
        c = cf.ew(coracle.MUL, a, b)
        sh = cf.split(c, coef, t, m)
        return cf.recombine([sh[j] for j in range(2 * t + 1)], lam)

    res = {}
    for label, threads in (('1core', 1), ('allcores', cores)):
        coracle.set_threads(threads)
        best = None
        for _ in range(2):
            t0 = time.perf_counter()
            y = one_pass()
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        res[label] = 3 * n / best
    coracle.set_threads(1)
    port_sample = (f'full workload: n={n} P61 elements x (modmul + split m={m},t={t} + recombine k={2*t+1}), '
                   f'oracle/fforacle.c with OpenMP ({cores} threads), best of 2')
    host_cores = os.cpu_count() or cores
    out = {'value': round(res['allcores'], 1), 'unit': 'field-ops/s', 'cores': cores, 'host_cores': host_cores, 'kind': 'port',
           'sample': port_sample, 'value_1core': round(res['1core'], 1)}
    ref_root = next((r_ for r_ in (os.path.join(ROOT, '_refstage'), '/root/reference') if os.path.isdir(os.path.join(r_, 'mpyc'))), None)
    r = None
    if ref_root is not None:
        procs = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
        n_one, n_each = 2_000_000, 400_000
        # (its own bounded process group: a pool of `procs` forked workers that loses one would never return)
        env = dict(os.environ, PYTHONPATH=os.pathsep.join([ROOT, ref_root]))
        got = run_bounded([sys.executable, os.path.join(ROOT, 'oracle', 'refbaseline.py'), ref_root, str(n_one), str(n_each), str(procs)],
                          env, 120.0, cwd=ROOT)
        line = None if got is None else next((ln for ln in got[1].splitlines() if ln.startswith('REFBASELINE ')), None)
        if line is not None:
            r = json.loads(line[len('REFBASELINE '):])
        if r is None or 'one_core' not in r:
            out['reference_error'] = 'the reference did not finish within its limit' if got is None else (got[1] + got[2])[-300:]
            r = None
    if r is not None:
        allc = r.get('all_cores') or {'field_ops_per_s': r['one_core']['field_ops_per_s'], 'n_total': n_one, 'wall_s': 0.0}
        procs = r.get('procs', procs)
        out = {'value': round(allc['field_ops_per_s'], 1), 'unit': 'field-ops/s', 'cores': procs, 'procs': procs,
               'host_cores': host_cores, 'kind': 'reference',
               'sample': f'lschoe/mpyc itself (FiniteFieldArray.__mul__ + thresha.np_random_split m={m},t={t} with live '
                         f'secrets.randbelow + thresha.np_recombine k={2*t+1}) over GF(2^61-1): {procs} processes x '
                         f'{n_each} elements = {allc["n_total"]} elements in {allc["wall_s"]:.2f} s wall; 1 core: '
                         f'{n_one} elements',
               'value_1core': round(r['one_core']['field_ops_per_s'], 1),
               'reference_1core_stages': {k_: round(v_, 1) for k_, v_ in r['one_core'].items() if k_.endswith('_per_s')},
               'process_count_probe': r.get('probe'),
               'port_value': round(res['allcores'], 1), 'port_value_1core': round(res['1core'], 1), 'port_cores': cores,
               'port_sample': port_sample}
    return out


def api_leg(n_full, parties_on_gpus=False):
    """The path THROUGH the reference's public API: tests/api_program.py -- an ordinary MPyC
    program, `mpc.output(a * b)` on SecFld(2^61-1) arrays, i.e. Runtime.np_multiply -> _reshare -> output
    (runtime.py:1096-1141, 603-689, 513-600) -- run as party processes under mpyc_amd.install(), and on the unmodified
    reference beside it (host cores).  Needs an importable mpyc (the staged copy _refstage/ on the GPU box).

    Per configuration: elements/s = n * multiplications / median wall time of one repetition at party 0 (inputs
    already shared; a repetition ends with the opened result on the device and a device synchronisation), and
    gpu_busy_frac = summed GPU time of all libffgpu calls of party 0 (one event pair per call, ffgpu_busy_ms) / wall
    time of the timed repetitions.  m = 1 is the runtime's own overhead + the kernels; m = 3 adds the reference's
    pickle + asyncio TCP transport between three local party processes (out of scope for the engine, SURVEY 8e:
    "party networking stays on the host") which then dominates: see `note`."""
    ref_root = next((r_ for r_ in (os.path.join(ROOT, '_refstage'), '/root/reference')
                     if os.path.isdir(os.path.join(r_, 'mpyc'))), None)
    if ref_root is None:
        return {'skipped': 'no importable mpyc checkout (stage one with tools/stage_reference.sh)'}
    import statistics
    import subprocess
    prog = os.path.join(ROOT, 'tests', 'api_program.py')

    def run(mode, n, parties, reps, warmup, chain=1, timeout=None, ipc_wire=False):
        timeout = LEG_BUDGET['leg_timeout'] if timeout is None else timeout
        env = dict(os.environ)
        env['MPYC_AMD_IPC_WIRE'] = '1' if ipc_wire else '0'
        env['PYTHONPATH'] = os.pathsep.join([os.path.join(ROOT, 'tests'), ROOT, ref_root])
        for k_ in ('MPYC_GPU', 'API_SEED', 'API_DIGEST', 'API_CPROFILE', 'RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'LOCAL_WORLD_SIZE',
                   'GROUP_RANK', 'MASTER_ADDR', 'MASTER_PORT', 'TORCHELASTIC_RUN_ID'):
            env.pop(k_, None)
        env['MPYC_AMD_DEVICE'] = 'party' if parties_on_gpus else 'current'
        env.update(API_MODE=mode, API_N=str(n), API_REPS=str(reps), API_WARMUP=str(warmup), API_CHAIN=str(chain))
        cmd = [sys.executable, prog, '--no-log'] + ([f'-M{parties}', '-B', str(free_base_port(parties))] if parties > 1 else [])
        t0 = time.perf_counter()
        if budget_left() < 3:
            return {'skipped': 'time budget of the run spent (bench.py --full runs every leg)'}
        r = run_bounded(cmd, env, timeout)
        if r is None:
            return {'error': f'no result within {min(timeout, LEG_BUDGET["leg_timeout"]):.0f} s (process group ended)'}
        rc, so, se = r
        line = next((ln for ln in so.splitlines() if ln.startswith('API_RESULT ')), None)
        if rc != 0 or line is None:
            return {'error': (so + se)[-400:]}
        d = json.loads(line[len('API_RESULT '):])
        med = statistics.median(d['times_s'])
        out = {'n': n, 'parties': parties, 't': d['t'], 'multiplications_per_rep': chain, 'reps': reps,
               'ms_per_rep': round(med * 1e3, 4), 'elements_per_s': round(n * chain / med, 1),
               'input_sharing_s': round(d['input_s'], 3), 'process_wall_s': round(time.perf_counter() - t0, 2)}
        if d.get('gpu_busy_ms') is not None:
            out['gpu_busy_ms_per_rep'] = round(d['gpu_busy_ms'] / reps, 4)
            # busy time per repetition / MEDIAN repetition (the figure elements_per_s is quoted on); the mean over all
            # repetitions, which a single host hiccup (GC, a late coroutine) moves, is given beside it
            out['gpu_busy_frac'] = round(d['gpu_busy_ms'] / reps * 1e-3 / med, 4)
            out['gpu_busy_frac_mean'] = round(d['gpu_busy_ms'] * 1e-3 / sum(d['times_s']), 4)
            out['libffgpu_calls_per_rep'] = d['gpu_calls'] / reps
        if parties > 1:
            out['wire'] = 'device buffers by interprocess handle (MPYC_AMD_IPC_WIRE=1)' if d.get('ipc_wire') else 'limb bytes through the TCP mesh'
            out['bytes_sent_party0'] = d.get('bytes_sent')
            out['device_party0'] = d.get('device')
        return out

    res = {'workload': 'mpc.output(a * b) on SecFld(GF(2^61-1)) arrays through the unmodified mpyc runtime under '
                       'mpyc_amd.install() (tests/api_program.py); reference = the same program without install()',
           'reference_root': os.path.basename(ref_root)}
    if parties_on_gpus:
        res = {'workload': 'mpc.output(a * b [* b ...]) on SecFld(GF(2^61-1)) arrays of 10^7 elements, THREE party processes on '
                           'GPUs 0, 1, 2 of this node (MPYC_AMD_DEVICE=party), share rows exchanged as interprocess handles '
                           '(device-side wire, mpyc_amd/ipcwire.py: peer copies between the GPUs)',
               'm3_1e7_ipc': run('gpu', n_full, 3, 10, 2, ipc_wire=True, timeout=120)}
        if 'error' not in res['m3_1e7_ipc']:         # (never measured across GPUs: a failure must cost the line two minutes, not ten)
            res['m3_1e7_chain8_ipc'] = run('gpu', n_full, 3, 5, 1, chain=8, ipc_wire=True, timeout=120)
        return res
    # (order: the legs the compact line quotes first -- the budget of a default run may end the list early)
    res['m1_1e7'] = run('gpu', n_full, 1, 20, 3)
    res['m1_1e8'] = run('gpu', 10 * n_full, 1, 5, 2)
    # three local parties with the device-side wire: share rows cross between the party processes as interprocess handles
    # of the device buffers (mpyc_amd/finfields.py _array_from_ipc), not as bytes through TCP
    res['m3_1e7_ipc'] = run('gpu', n_full, 3, 10, 2, ipc_wire=True)
    res['reference_m1_1e6'] = run('ref', n_full // 10, 1, 2, 0)
    res['m1_1e7_chain8'] = run('gpu', n_full, 1, 10, 2, chain=8)
    res['m3_1e7'] = run('gpu', n_full, 3, 3, 1)
    res['m3_1e6'] = run('gpu', n_full // 10, 3, 5, 1)
    res['m3_1e7_chain8_ipc'] = run('gpu', n_full, 3, 5, 1, chain=8, ipc_wire=True)
    res['m3_1e6_ipc'] = run('gpu', n_full // 10, 3, 10, 2, ipc_wire=True)
    res['reference_m3_1e6'] = run('ref', n_full // 10, 3, 1, 0)
    # the same runtime one protocol up: secure FIXED-POINT products (np_multiply + np_trunc -- random bits from PRSS, a masked
    # opening, integer arithmetic on `.value` that the device-resident views of install() keep on the GPU), SecFxp(32)
    def run_fxp(mode, n_, parties, timeout=None, prf=None, ipc=False):
        timeout = LEG_BUDGET['leg_timeout'] if timeout is None else timeout
        env = dict(os.environ)
        env['PYTHONPATH'] = os.pathsep.join([os.path.join(ROOT, 'tests'), ROOT, ref_root])
        for k_ in ('MPYC_GPU', 'RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'FXP_SEED', 'FXP_DIGEST', 'MPYC_AMD_PRSS_PRF'):
            env.pop(k_, None)
        # (one party: six repetitions -- the first two or three still pay allocator growth and clock ramp: 54 / 45 / 26 ms seen where
        # the steady state is 7-10)
        env.update(FXP_MODE=mode, FXP_N=str(n_), FXP_REPS=('6' if parties == 1 else '3') if mode != 'ref' else '1', MPYC_AMD_IPC_WIRE='1' if ipc else '0')
        if prf:
            env['MPYC_AMD_PRSS_PRF'] = prf
        cmd = [sys.executable, os.path.join(ROOT, 'tests', 'fxp_program.py'), '--no-log'] + \
              ([f'-M{parties}', '-B', str(free_base_port(parties))] if parties > 1 else [])
        t0 = time.perf_counter()
        if budget_left() < 3:
            return {'skipped': 'time budget of the run spent (bench.py --full runs every leg)'}
        r = run_bounded(cmd, env, timeout)
        if r is None:
            return {'error': f'no result within {min(timeout, LEG_BUDGET["leg_timeout"]):.0f} s (process group ended)'}
        rc, so, se = r
        line = next((ln for ln in so.splitlines() if ln.startswith('FXP_RESULT ')), None)
        if rc != 0 or line is None:
            return {'error': (so + se)[-300:]}
        d = json.loads(line[len('FXP_RESULT '):])
        secs = d['s_per_product_and_opening']
        return {'n': n_, 'parties': parties, 'prss_prf': d['prss_prf'], 's_per_product_and_opening': round(secs, 5),
                # the secure product alone (np_multiply + np_trunc: PRSS bits and masks, one masked opening; the share stays on the
                # device) -- the opening after it ends in the REFERENCE's conversion of n field elements to Python floats
                # (sectypes.py:1426-1447: np.vectorize of a Python lambda), host-bound whatever computed them
                's_per_product': round(d.get('s_per_product', float('nan')), 5),
                's_per_product_reps': [round(v, 5) for v in d.get('times_product_s', [])],       # (the value above is their minimum)
                'elements_per_s': round(n_ / secs, 1), 'products_per_s': round(n_ / d['s_per_product'], 1) if d.get('s_per_product') else None,
                # elements off by 2^48: the REFERENCE's own np_trunc mask for array types is f bits short (runtime.py:852; about
                # one element in 10^6; reproduced bit for bit: tests/test_fxp_path.py) -- counted, and excluded from the error
                'outliers_reference_trunc_mask': d['outliers_reference_trunc_mask'],
                'max_abs_error': d['max_abs_error_without_outliers'], 'process_wall_s': round(time.perf_counter() - t0, 2)}
    res['fxp_m1_1e6'] = run_fxp('gpu', n_full // 10, 1)
    res['fxp_m1_1e6_chacha'] = run_fxp('gpu', n_full // 10, 1, prf='chacha')
    res['fxp_m3_1e6_ipc'] = run_fxp('gpu', n_full // 10, 3, ipc=True)
    res['fxp_m3_1e6_ipc_chacha'] = run_fxp('gpu', n_full // 10, 3, prf='chacha', ipc=True)
    res['reference_fxp_m1_2e4'] = run_fxp('ref', n_full // 500, 1)
    if isinstance(res['reference_fxp_m1_2e4'], dict) and 'elements_per_s' in res['reference_fxp_m1_2e4']:
        res['reference_fxp_m1_2e4']['note'] = ('the short-mask outlier is NOT observable at this size (expected 0.02 per run; the '
                                               'reference needs ~150 s for the 10^6 elements that show one): the digest-level test runs '
                                               'both sides on keys that produce one at n = 10^5')
    if 'elements_per_s' in res['fxp_m1_1e6'] and 'elements_per_s' in res['reference_fxp_m1_2e4']:
        res['fxp_vs_reference_m1'] = round(res['fxp_m1_1e6']['elements_per_s'] / res['reference_fxp_m1_2e4']['elements_per_s'], 1)
    if 'elements_per_s' in res['fxp_m1_1e6_chacha'] and 'elements_per_s' in res['reference_fxp_m1_2e4']:
        res['fxp_chacha_vs_reference_m1'] = round(res['fxp_m1_1e6_chacha']['elements_per_s'] / res['reference_fxp_m1_2e4']['elements_per_s'], 1)
    head = res['m1_1e7']
    if 'elements_per_s' in head:
        res['elements_per_s'] = head['elements_per_s']
        res['gpu_busy_frac'] = head.get('gpu_busy_frac')
        res['gpu_busy_frac_1e8'] = res['m1_1e8'].get('gpu_busy_frac')
        ref = res['reference_m1_1e6']
        if 'elements_per_s' in ref:
            res['vs_reference_m1'] = round(head['elements_per_s'] / ref['elements_per_s'], 1)
        if 'elements_per_s' in res['m3_1e6'] and 'elements_per_s' in res['reference_m3_1e6']:
            res['vs_reference_m3_1e6'] = round(res['m3_1e6']['elements_per_s'] / res['reference_m3_1e6']['elements_per_s'], 1)
            if 'elements_per_s' in res['m3_1e6_ipc']:
                res['vs_reference_m3_1e6_ipc'] = round(res['m3_1e6_ipc']['elements_per_s'] / res['reference_m3_1e6']['elements_per_s'], 1)
        res['gpu_busy_frac_m3_1e7_ipc'] = res['m3_1e7_ipc'].get('gpu_busy_frac')
    res['note'] = ('m=1: one repetition = np_multiply + output coroutines of the reference runtime (host, ~0.1 ms) around two '
                   'kernels (product, recombination); the GPU-busy share grows with n (1e8: kernels dominate) and with the number '
                   'of multiplications in flight (chain8: launches overlap the host). m=3: every gate moves 2 x n x 8 B out of and '
                   'into each party through pickle + asyncio TCP of the reference (asyncoro.py:54-106), ~0.8 s per gate at n=1e7 '
                   'against ~0.4 ms of kernels -- the engine is idle; see profiles/r03_api_path.md')
    return res


def list_path_leg():
    """BASELINE.json configs[0] at its stated size: thresha.random_split + recombine on the LIST path, m = 3, t = 1, GF(2^61-1),
    10^4 secrets (thresha.py:23-44, 88-116) -- tests/list_path_program.py: the mirror under install() (device path from
    mpyc_amd.list_path_min secrets on) beside the reference's own functions in one process; parity on replayed draws
    (element by element, incl. the reference's un-reduced sums), then both timed with live randomness.  A host-bound
    row by nature: 10^4 Python integers in and out per call; it is here for coverage of the config, not for the roofline."""
    ref_root = next((r_ for r_ in (os.path.join(ROOT, '_refstage'), '/root/reference')
                     if os.path.isdir(os.path.join(r_, 'mpyc'))), None)
    if ref_root is None:
        return {'skipped': 'no importable mpyc checkout (stage one with tools/stage_reference.sh)'}
    import subprocess
    env = dict(os.environ)
    env['PYTHONPATH'] = os.pathsep.join([os.path.join(ROOT, 'tests'), ROOT, ref_root])
    for k_ in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MPYC_AMD_CPUCTX'):
        env.pop(k_, None)
    env.update(LP_MODE='gpu', LP_N='10000', LP_M='3', LP_T='1', LP_PRIME=str(P61), LP_SEED='5', LP_REPS='5')
    r = run_bounded([sys.executable, os.path.join(ROOT, 'tests', 'list_path_program.py')], env, LEG_BUDGET['leg_timeout'])
    if r is None:
        return {'error': 'no result within the limit'}
    rc, so, se = r
    line = next((ln for ln in so.splitlines() if ln.startswith('LIST_PATH_RESULT ')), None)
    if rc != 0 or line is None:
        return {'error': (so + se)[-400:]}
    d = json.loads(line[len('LIST_PATH_RESULT '):])
    return {'workload': 'configs[0]: thresha.random_split + recombine(k=2), list path, m=3, t=1, GF(2^61-1), 10^4 secrets',
            'name': 'list_path_p61_1e4_m3t1', 'n': d['n'], 'unit': 'secrets/s',
            'mirror_secrets_per_s': round(d['mirror_secrets_per_s'], 1), 'reference_secrets_per_s': round(d['reference_secrets_per_s'], 1),
            'mirror_ms': round(d['mirror_s'] * 1e3, 3), 'reference_ms': round(d['reference_s'] * 1e3, 3),
            'parity': 'element by element vs the reference on replayed draws (%d point sets)' % d['subsets_checked'],
            'split_digest': d['split_digest'], 'opened_digest': d['opened_digest'], 'list_path_min': d['list_path_min'],
            'bound': 'host (Python integers in and out; the kernels take microseconds)'}


def oracle_sample_check(modulus, a, b, coef, shares, y, t, m, lam, seed=7, count=16384):
    """In-run parity guard against the C ORACLE (oracle/fforacle.c; checker only, outside every timed region): for the
    first and last 4096 elements and `count` random positions, shares == split(a*b; coef) row by row and
    y == recombine(2t+1 rows) == a*b, all computed by the oracle from the inputs at those positions."""
    from oracle import coracle
    n = a.n
    g = np.random.default_rng(seed)
    idx = np.unique(np.concatenate([np.arange(min(4096, n)), np.arange(max(0, n - 4096), n),
                                    g.integers(0, n, size=count)])).astype(np.int64)
    tidx = torch.from_numpy(idx).to(a.t.device)

    def take(d):
        return np.ascontiguousarray(d.t.index_select(0, tidx).cpu().numpy()).view(np.uint64)
    cf = coracle.CField(modulus)
    A, B = take(a), take(b)
    C = np.stack([take(coef.row(j)) for j in range(t)])
    prod = cf.ew(coracle.MUL, A, B)
    sh = cf.split(prod, C, t, m)
    rec = cf.recombine([sh[j] for j in range(2 * t + 1)], lam)
    ok = bool((rec == prod).all()) and bool((take(y) == rec).all())
    for i in range(m):
        ok = ok and bool((take(shares.row(i)) == sh[i]).all())
    return ok


def u128_rows(rows, n, device, gen):
    """(rows, n, 2) int64 limb pairs: uniform canonical elements of GF(2^128 - 173)."""
    x = torch.randint(-2**63, 2**63 - 1, (rows, n, 2), dtype=torch.int64, device=device, generator=gen)
    # (hi, lo) >= p only if hi == 2^64-1 and lo >= 2^64-173: fold those few
    bad = (x[..., 1] == -1) & (x[..., 0] < 0) & (x[..., 0] >= -173)
    x[..., 0] = torch.where(bad, x[..., 0] + 173, x[..., 0])
    x[..., 1] = torch.where(bad, torch.zeros_like(x[..., 1]), x[..., 1])
    return x


def multi_gpu_leg(dist, rank, world, local_rank, backend, n, steps, warmup, lagrange):
    """configs[3] on N GPUs (every rank runs this): 128-bit prime 2^128-173, m = 7, t = 3, n gates per GPU.

    gate_sharded    the production layout: gates sharded by element, every rank runs fused mul+split and the
                    recombination from 2t+1 = 7 rows on its own shard -- no collective.
    party_major_*   the one layout with an exchange step: share row j lives whole on rank j % N (the GPUs stand in
                    for the parties); before recombining, every rank needs its column range of all 7 rows.
                    `all_to_all`: multigpu.exchange_party_major (batched point-to-point of column slices, 1/N of
                    the all-gather traffic); `allgather`: multigpu.PartyMajorGather (ONE all_gather_into_tensor of
                    whole rows -- the north star's wording).  Both are followed by the same k = 7 recombination of
                    the rank's own column range.  Rows have N*n elements (weak scaling: n secrets per GPU).
    Times are wall clock between barriers, MAX over ranks; `exchange_share` = the exchange timed alone / the step."""
    from mpyc_amd.engine import DevArray, FieldContext
    from mpyc_amd import multigpu
    P128 = 2**128 - 173
    t, m = 3, 7
    k = 2 * t + 1
    eb = 16
    ctx = FieldContext(P128, device=local_rank)
    dev = ctx.torch_device
    lam = lagrange(P128, range(1, k + 1))
    red_dev = dev if backend == 'nccl' else 'cpu'

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(fn):
        for _ in range(warmup):
            fn()
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        barrier()
        dt = time.perf_counter() - t0
        if dist is not None:
            tt = torch.tensor([dt], dtype=torch.float64, device=red_dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        return dt / steps * 1e3          # ms per step

    fails = []          # parity checks that failed on THIS rank: reported, all-reduced (MIN) into the line, never raised --
    #                     a rank that leaves here would strand the others in their next collective
    res = {'config': {'workload': 'configs[3]: GF(2^128-173) (two limbs), m=7, t=3, gate = modmul + np_random_split + '
                                  'np_recombine(k=7)', 'n_per_gpu': n, 'n_gpus': world, 'backend': backend,
                      'steps': steps, 'warmup': warmup}}
    gen = torch.Generator(device=dev)

    # ---- element-sharded gate: no collective -----------------------------------------------------------------
    gen.manual_seed(977 + rank)
    sets = []
    for _ in range(2):
        ab = u128_rows(2 + t, n, dev, gen)
        coef = ctx.empty_matrix(t, n)
        for j in range(t):
            coef.row(j).t.copy_(ab[2 + j])
        shares = ctx.empty_matrix(m, n)
        y = ctx.empty(n)
        sets.append({'a': DevArray(ctx, ab[0].contiguous(), n), 'b': DevArray(ctx, ab[1].contiguous(), n), 'coef': coef,
                     'shares': shares, 'y': y, 'rec': ctx.recombine_plan([shares.row(j) for j in range(k)], lam, y)})
        del ab
    it = [0]

    def gate():
        s_ = sets[it[0] % len(sets)]
        it[0] += 1
        ctx.split(s_['a'], s_['coef'], t, m, out=s_['shares'], mul_by=s_['b'])
        s_['rec']()
    ms = timed(gate)
    torch.cuda.synchronize()
    s0_ = sets[(it[0] - 1) % len(sets)]
    if not oracle_sample_check(P128, s0_['a'], s0_['b'], s0_['coef'], s0_['shares'], s0_['y'], t, m, lam):
        fails.append('gate_sharded: P128 gate differs from the oracle on the sampled positions')
    bpu = (2 + t + m) * eb + (k + 1) * eb
    res['gate_sharded'] = {'ms_per_step': round(ms, 5), 'gates_per_s': round(n * world / (ms * 1e-3), 1),
                           'algorithmic_bytes_per_gate': bpu, 'GBps_per_gpu': round(bpu * n / (ms * 1e-3) / 1e9, 1),
                           'frac_of_hbm_peak': round(bpu * n / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), 'collective': None}
    del sets
    torch.cuda.empty_cache()

    # ---- party-major rows: exchange + recombination of the own column range -------------------------------
    ntot = n * world
    lo, hi = multigpu.shard_range(ntot, rank, world)

    def row_slice(j, r):
        """columns of rank r of share row j: reproducible anywhere from (j, r) -- lets every rank check what it receives"""
        g = torch.Generator(device=dev)
        g.manual_seed(1_000_003 * (j + 1) + r)
        return u128_rows(1, multigpu.shard_range(ntot, r, world)[1] - multigpu.shard_range(ntot, r, world)[0], dev, g)[0]

    template = torch.empty((0, 2), dtype=torch.int64, device=dev)
    local = {j: torch.cat([row_slice(j, r) for r in range(world)]) for j in range(k) if multigpu.row_owner(j, world) == rank}
    want = [row_slice(j, rank) for j in range(k)]
    recv = [torch.empty((hi - lo, 2), dtype=torch.int64, device=dev) for _ in range(k)]
    y = ctx.empty(hi - lo)
    row_ids = list(range(k))
    got = multigpu.exchange_party_major(local, row_ids, ntot, template=template, recv=recv)
    torch.cuda.synchronize()
    for j in range(k):
        if not torch.equal(got[j], want[j]):
            fails.append(f'party_major: exchanged slice of row {j} differs on rank {rank}')
    y_want = ctx.recombine([DevArray(ctx, w_, hi - lo) for w_ in want], lam)

    def a2a_exchange():
        return multigpu.exchange_party_major(local, row_ids, ntot, template=template, recv=recv)

    def a2a_step():
        sl = a2a_exchange()
        ctx.recombine([DevArray(ctx, s_, hi - lo) for s_ in sl], lam, out=y)
    ms_step = timed(a2a_step)
    torch.cuda.synchronize()
    if not torch.equal(y.t, y_want.t):
        fails.append('party_major_all_to_all: recombination differs from the local one')
    ms_x = timed(a2a_exchange)
    owned = len(local)
    sent = owned * (ntot - (hi - lo)) * eb                         # bytes this rank sends to its peers per step
    rcvd = (k - owned) * (hi - lo) * eb
    res['party_major_all_to_all'] = {
        'ms_per_step': round(ms_step, 5), 'ms_exchange_alone': round(ms_x, 5),
        'exchange_share': round(ms_x / ms_step, 4), 'secrets_per_s': round(ntot / (ms_step * 1e-3), 1),
        'rank0_bytes_sent': sent, 'rank0_bytes_received': rcvd,
        'rank0_exchange_GBps': round((sent + rcvd) / (ms_x * 1e-3) / 1e9, 1) if world > 1 else 0.0,
        'collective': 'batched isend/irecv of column slices (exchange_party_major)'}
    # the same step with the exchange PIPELINED against the recombination: 4 column chunks, transfers of chunk c+1 in
    # flight while the kernel of chunk c runs (multigpu.recombine_party_major(chunks=4))
    y.t.zero_()

    def a2a_pipelined():
        multigpu.recombine_party_major(ctx, local, row_ids, lam, ntot, template=template, chunks=4, out=y, recv=recv)
    ms_pipe = timed(a2a_pipelined)
    torch.cuda.synchronize()
    if not torch.equal(y.t, y_want.t):
        fails.append('party_major_all_to_all_pipelined: recombination differs from the local one')
    res['party_major_all_to_all_pipelined'] = {
        'ms_per_step': round(ms_pipe, 5), 'chunks': 4, 'secrets_per_s': round(ntot / (ms_pipe * 1e-3), 1),
        'vs_sequential': round(ms_step / ms_pipe, 3),
        'collective': 'batched isend/irecv per column chunk, overlapped with k_recombine of the previous chunk'}
    del recv, got

    pg = multigpu.PartyMajorGather(k, ntot, template)
    for j, row in local.items():
        pg.block_row(j).copy_(row)
    del local
    torch.cuda.empty_cache()
    rows_view = None

    def ag_step():
        pg.gather()
        ctx.recombine([DevArray(ctx, pg.row(j)[lo:hi], hi - lo) for j in range(k)], lam, out=y)
    y.t.zero_()
    ms_step = timed(ag_step)
    torch.cuda.synchronize()
    if not torch.equal(y.t, y_want.t):
        fails.append('party_major_allgather: recombination differs from the local one')
    ms_x = timed(pg.gather)
    res['party_major_allgather'] = {
        'ms_per_step': round(ms_step, 5), 'ms_exchange_alone': round(ms_x, 5),
        'exchange_share': round(ms_x / ms_step, 4), 'secrets_per_s': round(ntot / (ms_step * 1e-3), 1),
        'bytes_received_per_rank': pg.bytes_received,
        'exchange_GBps_per_rank': round(pg.bytes_received / (ms_x * 1e-3) / 1e9, 1) if world > 1 else 0.0,
        'collective': 'all_gather_into_tensor of whole rows (PartyMajorGather)'}
    # every rank's checks, as ONE flag: MIN over ranks (1 = every check passed everywhere)
    flag = torch.tensor([0 if fails else 1], dtype=torch.int32, device=red_dev)
    if dist is not None:
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    res['parity'] = {'passed_on_every_rank': bool(int(flag.item())), 'failures_on_rank_0': fails,
                     'checks': ['gate vs oracle/fforacle.c on sampled positions', 'every exchanged row slice vs its seed',
                                'recombination after all-to-all / pipelined all-to-all / all-gather vs the local recombination']}
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--n', type=int, default=int(os.environ.get('FFGPU_BENCH_N', 10_000_000)), help='elements per GPU')
    ap.add_argument('--sets', type=int, default=4, help='rotating buffer sets')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-api-leg', action='store_true', help='skip the API-level section (party processes under install())')
    ap.add_argument('--no-extras', action='store_true')
    ap.add_argument('--budget', type=float, default=float(os.environ.get('FFGPU_BENCH_BUDGET', '240')),
                    help='seconds the sections that run OTHER programs (reference baseline, API-level legs) may take together; '
                         'legs that no longer fit are reported as skipped (default 240; each leg also has a 60 s limit of its own)')
    ap.add_argument('--full', action='store_true', help='no budget, 300 s per leg: the evidence passes')
    ap.add_argument('--no-multi-gpu-leg', action='store_true', help='skip the configs[3] / party-major section')
    ap.add_argument('--parties-on-gpus', action='store_true',
                    default=os.environ.get('FFGPU_BENCH_PARTIES_ON_GPUS', '0') == '1',
                    help='N >= 3: also run three MPyC parties on three GPUs over the device-side wire (opt-in; never run on >1 GPU)')
    ap.add_argument('--layout', choices=('element', 'party-major'), default='element',
                    help="'party-major': only the configs[3] section (gate sharded + party-major exchange), more steps")
    args = ap.parse_args()

    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
        # invoked as plain `python bench.py --gpus N`: re-launch under torch.distributed.run, one rank per GPU
        # (rendezvous on 127.0.0.1; the container hostname may not resolve)
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(('127.0.0.1', 0))
            port = sk.getsockname()[1]
        # (`--n` is an ambiguous abbreviation for torch.distributed.run's own parser: it travels in the environment)
        fwd = ['--gpus', str(args.gpus), '--steps', str(args.steps), '--warmup', str(args.warmup), '--sets', str(args.sets),
               '--layout', args.layout, '--budget', str(args.budget)]
        fwd += [f_ for f_, on in (('--no-cpu-baseline', args.no_cpu_baseline), ('--no-extras', args.no_extras),
                                  ('--no-api-leg', args.no_api_leg), ('--no-multi-gpu-leg', args.no_multi_gpu_leg),
                                  ('--parties-on-gpus', args.parties_on_gpus), ('--full', args.full)) if on]
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={args.gpus}',
               '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + fwd
        raise SystemExit(subprocess.call(cmd, env=dict(os.environ, FFGPU_BENCH_N=str(args.n))))
    if world != args.gpus:
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU')
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a GPU: the hot path has no CPU fallback')
    # validation hooks (not used by the driver): run the N>1 control flow on a 1-GPU box
    force_dev = os.environ.get('FFGPU_BENCH_DEVICE')
    if force_dev is not None:
        local_rank = int(force_dev)
    # (ranks forced onto ONE device cannot form an RCCL communicator -- "Duplicate GPU detected" -- so gloo is the default then)
    backend = os.environ.get('FFGPU_BENCH_BACKEND', 'gloo' if (force_dev is not None and world > 1) else 'nccl')
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist_
        dist = dist_
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))
        else:
            dist.init_process_group(backend)

    from mpyc_amd.engine import FieldContext
    from mpyc_amd import finfields as gff, thresha as gth

    def lagrange(modulus, xs):
        """Recombination vector at 0 from the product's own host code (thresha.py:67-85 mirror)."""
        return list(gth._recombination_vector(gff.GF(modulus), tuple(xs), 0))

    n, t, m = args.n, 1, 3
    k = 2 * t + 1
    ctx = FieldContext(P61, device=local_rank)
    gen = torch.Generator(device=ctx.torch_device)
    gen.manual_seed(20260925 + rank)
    sets = [StepData(ctx, n, t, m, gen) for _ in range(args.sets)]
    lam = lagrange(P61, range(1, k + 1))

    def f_mul(s):
        ctx.mul(s.a, s.b, out=s.c)

    def f_split(s):
        ctx.split(s.c, s.coef, t, m, out=s.shares)

    for s in sets:
        s.rec = ctx.recombine_plan([s.shares.row(j) for j in range(k)], lam, s.y)

    def f_rec(s):
        s.rec()

    def f_fused(s):
        ctx.split(s.a, s.coef, t, m, out=s.shares, mul_by=s.b)

    def step(i):
        s = sets[i % len(sets)]
        f_fused(s)          # c = a*b in registers -> shares  (never written)
        f_rec(s)            # shares -> y

    def step_unfused(i):
        s = sets[i % len(sets)]
        f_mul(s)
        f_split(s)
        f_rec(s)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i)
    # parity guard inside the bench: recombining the fresh shares gives back a*b
    torch.cuda.synchronize()
    s0 = sets[(args.warmup - 1) % len(sets)] if args.warmup else None
    if s0 is not None:
        if not oracle_sample_check(P61, s0.a, s0.b, s0.coef, s0.shares, s0.y, t, m, lam):
            raise SystemExit('bench parity check failed: the step differs from the oracle on the sampled positions')

    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i)
    barrier()
    elapsed = time.perf_counter() - t0
    red_dev = ctx.torch_device if backend == 'nccl' else 'cpu'
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    # the same work as three separate kernels (c materialised), for reference
    for i in range(min(args.warmup, 3)):
        step_unfused(i)
    barrier()
    t1 = time.perf_counter()
    for i in range(args.steps):
        step_unfused(i)
    barrier()
    elapsed_unfused = time.perf_counter() - t1
    if dist is not None:
        tt = torch.tensor([elapsed_unfused], dtype=torch.float64, device=red_dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed_unfused = float(tt.item())

    devinfo = f'cuda:{local_rank} ({torch.cuda.get_device_name(local_rank)})'
    me = {'rank': rank, 'local_rank': local_rank, 'device': devinfo, 'pci_bus_id': ctx.pci_bus_id()}
    if dist is not None:
        gathered = [None] * world
        dist.all_gather_object(gathered, me)
    else:
        gathered = [me]
    try:
        rccl_version = '.'.join(str(v_) for v_ in torch.cuda.nccl.version())
    except Exception:          # noqa: BLE001 -- informational only
        rccl_version = None
    ops_total = 3.0 * n * world * args.steps
    value = ops_total / elapsed
    out = {
        'metric': 'field-ops/sec (modmul + share+recombine) on 10^7-elt SecFld array',
        'value': round(value, 1), 'unit': 'field-ops/s', 'n_gpus': world, 'steps': args.steps,
        'warmup': args.warmup, 'ms_per_step': round(elapsed / args.steps * 1e3, 5),
        'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'u64',
        'data': 'synthetic',
        'config': {'workload': 'configs[1]: SecFld(GF(2^61-1)) array, 10^7 elements per GPU; step = '
                               'modmul + np_random_split(m=3,t=1, coefficients supplied) + np_recombine(k=3); '
                               'modmul is fused into the share-generation kernel (product never written to HBM)',
                   'n_per_gpu': n, 'prime': '2^61-1', 'm': m, 't': t, 'k': k, 'field_ops_per_step': 3 * n,
                   'buffer_sets': args.sets, 'parallelism': f'element-sharded x{world}, no collective'},
        'distributed': {'backend': (backend if dist is not None else 'none (single process)'), 'world_size': world,
                        'collective_library': 'RCCL (torch.distributed nccl backend)' if dist is not None and backend == 'nccl' else 'none',
                        'rccl_version': rccl_version,
                        'distinct_devices': len({r_['pci_bus_id'] for r_ in gathered}),
                        'ranks': gathered},
    }

    invalid = []
    if dist is not None and backend == 'nccl' and out['distributed']['distinct_devices'] < world:
        # two RCCL ranks on one GPU: the aggregate would count that GPU twice -- say so in the line and fail the run
        invalid.append('invalid: ranks share a device')
        out['scaling'] = invalid[0]
    out['unfused'] = {'value': round(ops_total / elapsed_unfused, 1), 'unit': 'field-ops/s',
                      'ms_per_step': round(elapsed_unfused / args.steps * 1e3, 5),
                      'note': 'same step as three kernels (mul, split, recombine) with c written to HBM'}

    if args.layout == 'party-major':
        args.no_extras = True
        args.no_cpu_baseline = True

    if rank == 0 and not args.no_extras:
        eb = 8
        reps = 10
        kern = {}
        ms = time_launches(f_mul, sets, reps)
        kern['mul_p61'] = dict(roof(3 * eb * n, ms), kernel='k_ew2<PM64<false,true>,MUL>',
                               algorithmic_bytes_per_unit=3 * eb, units_per_s=round(n / (ms * 1e-3), 1))
        ms = time_launches(f_split, sets, reps)
        bpu = (1 + t + m) * eb
        kern['split_p61_m3t1'] = dict(roof(bpu * n, ms), kernel='k_split<PM64<false,true>,1>',
                                      algorithmic_bytes_per_unit=bpu, units_per_s=round(n / (ms * 1e-3), 1))
        ms = time_launches(f_rec, sets, reps)
        bpu = (k + 1) * eb
        kern['recombine_p61_k3'] = dict(roof(bpu * n, ms), kernel='k_recombine<PM64<false,true>,3>',
                                        algorithmic_bytes_per_unit=bpu, units_per_s=round(n / (ms * 1e-3), 1))
        for rounds in (20, 8):
            ms = time_launches(lambda s: ctx.split_rng(s.c, t, m, key=bytes(range(32)), nonce=3, rounds=rounds,
                                                       out=s.shares), sets, reps)
            bpu = (1 + m) * eb
            kern[f'split_rng_p61_m3t1_chacha{rounds}'] = dict(roof(bpu * n, ms), algorithmic_bytes_per_unit=bpu,
                                                              units_per_s=round(n / (ms * 1e-3), 1))
        # fused local product + share generation (c never written)
        ms = time_launches(f_fused, sets, reps)
        bpu = (2 + t + m) * eb
        kern['mul_split_fused_p61_m3t1'] = dict(roof(bpu * n, ms), kernel='k_split<PM64<false,true>,T=1,fused mul,nt>',
                                                algorithmic_bytes_per_unit=bpu,
                                                units_per_s=round(n / (ms * 1e-3), 1))
        # achievable-bandwidth yardstick: the library's streaming copy, 80 MB blocks rotating
        ms_copy = time_launches(lambda s: ctx.copy(s.a.t, s.c.t), sets, reps)
        kern['device_copy'] = dict(roof(2 * eb * n, ms_copy), kernel='k_copy16 (80 MB -> 80 MB, rotating sets)')
        def optional_measurements():
            """Everything beyond the step's own kernels; a failure here must not cost the headline line."""
            from mpyc_amd import gfpx as ggx, protocols
            lap('before_extras')
            # A gate INSIDE a chain of multiplications (production mode): the k = 3 sub-share rows received in the
            # previous gate are recombined in registers, squared and re-shared with the device CSPRNG in ONE kernel
            # (ffgpu_gate_rng): 3 reads + 3 writes per element, all three stages of the headline step.
            st_chain = ctx.rng_state()
            outs = [ctx.empty_matrix(m, n) for _ in range(2)]
            chain_sets = [(sets[i].shares, outs[i % 2]) for i in range(len(sets))]
            for rounds in (20, 8):
                stc = ctx.rng_state(rounds=rounds)
                ms = time_launches(lambda s_: ctx.gate([s_[0].row(j) for j in range(k)], lam, None, None, t, m, state=stc,
                                                       out=s_[1]), chain_sets, reps)
                bpu = (k + m) * eb
                kern[f'chain_gate_p61_m3t1_chacha{rounds}'] = dict(roof(bpu * n, ms), algorithmic_bytes_per_unit=bpu,
                                                                   units_per_s=round(n / (ms * 1e-3), 1),
                                                                   field_ops_per_s=round(3 * n / (ms * 1e-3), 1))
            chk = ctx.gate([sets[0].shares.row(j) for j in range(k)], lam, None, None, t, m, state=st_chain, out=outs[0])
            y0 = sets[0].rec()
            if not torch.equal(ctx.recombine([chk.row(j) for j in range(k)], lam).t, ctx.mul(y0, y0).t):
                raise SystemExit('bench parity check failed for the chain gate')
            del outs, chain_sets
            lap('chain_gate')
            # second-tier element-wise ops (finfields.py:1278-1281,1424-1458): batched inverse, sqrt = pow by (p+1)/4
            ms = time_launches(lambda s: ctx.inv(s.a, out=s.c, check_zero=False), sets, 3)
            kern['inv_p61'] = dict(roof(2 * eb * n, ms), algorithmic_bytes_per_unit=2 * eb, bound_note='integer ALU',
                                   units_per_s=round(n / (ms * 1e-3), 1))
            ms = time_launches(lambda s: ctx.pow(s.a, (P61 + 1) // 4, out=s.c), sets, 3)
            kern['sqrt_p61'] = dict(roof(2 * eb * n, ms), algorithmic_bytes_per_unit=2 * eb, bound_note='integer ALU',
                                    units_per_s=round(n / (ms * 1e-3), 1))
            # (both are bound by integer VALU work, not by HBM: `bound`, the instruction counts and `valu_frac` are filled in by
            # annotate_valu from profiles/r05_valu.json and the issue rate measured in this run)
            lap('inv_sqrt')
            # PRSS (thresha.py:163-173) for one party of m = 7, t = 3: 20 subset keys x n x 28 B of SHAKE128 on 20 host
            # threads, squeezed / uploaded / combined in slices -- bound by the host sponge, the device part is hidden
            import itertools
            from mpyc_amd import finfields as gff, thresha as gth
            F61 = gff.GF(P61)
            keys7 = {S: bytes([sum(S) % 256]) * 16 + bytes(S) for S in itertools.combinations(range(7), 4) if 2 in S}
            prfs7 = {S: gth.PRF(kk_, F61.order) for S, kk_ in keys7.items()}
            npr = min(n, 2 * 10**6)
            gth.np_pseudorandom_share(F61, 7, 2, prfs7, b'warm', npr)
            torch.cuda.synchronize()
            t0_ = time.perf_counter()
            gth.np_pseudorandom_share(F61, 7, 2, prfs7, b'uci', npr)
            torch.cuda.synchronize()
            dt_ = time.perf_counter() - t0_
            lb_ = next(iter(prfs7.values())).byte_length
            kern['prss_share_p61_m7t3'] = {'ms_per_launch': round(dt_ * 1e3, 2), 'bound': 'host', 'unit': 'GB/s',
                                           'achieved': round(len(prfs7) * npr * lb_ / dt_ / 1e9, 2), 'frac': None,
                                           'units_per_s': round(npr / dt_, 1), 'n': npr, 'subset_keys': len(prfs7),
                                           'xof_bytes_per_draw': lb_,
                                           'note': 'PARITY mode (the reference PRF): achieved = SHAKE128 output bytes/s over all subset '
                                                   'keys (host threads, ffgpu_shake128_squeeze); upload and ffgpu_prss_combine overlap it'}
            lap('prss_parity')
            # the same shares in PRODUCTION mode (thresha.prss_prf = 'chacha'): one ChaCha stream per subset key expanded by
            # the lanes that consume the draws (ffgpu_prss_chacha) -- nothing crosses PCIe, 8 B written per share; VALU-bound
            # (20 draws of 24 keystream bytes per share at m = 7, t = 3: 7.5 ChaCha blocks), priced by annotate_valu
            keys3 = {S: bytes([sum(S) % 256]) * 16 + bytes(S) for S in itertools.combinations(range(3), 2) if 0 in S}
            prfs3 = {S: gth.PRF(kk_, F61.order) for S, kk_ in keys3.items()}
            prev_mode, prev_rounds, prev_allow8 = gth.prss_prf, gth.prss_rounds, gth.prss_allow_chacha8
            try:
                gth.prss_prf, gth.prss_allow_chacha8 = 'chacha', True          # (the ChaCha8 row is a measurement)
                for (mm_, ii_, pr_), rr_ in itertools.product(((7, 2, prfs7), (3, 0, prfs3)), (20, 8)):
                    gth.prss_rounds = rr_
                    for _ in range(3):          # (the host-bound parity-mode run above lets the clocks drop: ramp them up again)
                        gth.np_pseudorandom_share(F61, mm_, ii_, pr_, b'uci', n)
                    ms = time_launches(lambda s_: gth.np_pseudorandom_share(F61, mm_, ii_, pr_, b'uci', n), [0], 10)
                    tt_ = (mm_ - 1) // 2
                    kern[f'prss_share_p61_m{mm_}t{tt_}_chacha{rr_}'] = dict(
                        roof(eb * n, ms), algorithmic_bytes_per_unit=eb, units_per_s=round(n / (ms * 1e-3), 1), bound='valu',
                        subset_keys=len(pr_), keystream_bytes_per_share=len(pr_) * lb_,
                        note='PRODUCTION mode: ChaCha counter-mode PRF on the device, same sampling rule and combination as the '
                             'reference (thresha.py:163-173, 234-266); frac = HBM fraction of the 8 B written per share')
                # in-run guard (checker only, outside every timed region): the first and last 4096 shares of the m = 7 party against
                # the C oracle's restatement of the keystream layout
                from oracle import coracle as _co
                gth.prss_rounds = 20
                shr = gth.np_pseudorandom_share(F61, 7, 2, prfs7, b'uci', 8192 * 3 + 5)
                k40_ = [gth.prss_chacha_stream_key(pf_.key, b'uci') for pf_ in prfs7.values()]
                w_ = [int(gth._f_S_i(F61, 7, 2, S_)) for S_ in prfs7]
                ref_ = _co.prss_chacha(_co.CField(P61), k40_, 1, lb_, 0, 20, w_, 8192 * 3 + 5)
                if not (shr.device_array.to_numpy() == ref_).all():
                    raise SystemExit('bench parity check failed: production-mode PRSS shares differ from the oracle')
            finally:
                gth.prss_prf, gth.prss_rounds, gth.prss_allow_chacha8 = prev_mode, prev_rounds, prev_allow8
            lap('prss_chacha')
            # boundary handed HOST buffers (pinned): h2d of both operands + mulmod + d2h of the product, end to end
            # through ffgpu_h2d / ffgpu_mul / ffgpu_d2h.  Reported for DESIGN.md only -- never the headline value.
            from mpyc_amd import _ffi
            ha, hb, hc = (torch.empty(n, dtype=torch.int64).pin_memory() for _ in range(3))
            ha.copy_(sets[0].a.t.cpu())
            hb.copy_(sets[0].b.t.cpu())
            L, h, st = ctx._L, ctx._h, ctx._stream()

            def host_mul():
                _ffi.check(L.ffgpu_h2d(h, sets[0].a.ptr, ha.data_ptr(), n * eb, st), 'h2d')
                _ffi.check(L.ffgpu_h2d(h, sets[0].b.ptr, hb.data_ptr(), n * eb, st), 'h2d')
                ctx.mul(sets[0].a, sets[0].b, out=sets[0].c)
                _ffi.check(L.ffgpu_d2h(h, hc.data_ptr(), sets[0].c.ptr, n * eb, st), 'd2h')
            ms = time_launches(lambda s_: host_mul(), [0], 3)
            kern['mul_p61_pcie_inclusive'] = {'ms_per_launch': round(ms, 4), 'unit': 'GB/s', 'bound': 'pcie',
                                              'achieved': round(3 * eb * n / (ms * 1e-3) / 1e9, 1), 'frac': None,
                                              'units_per_s': round(n / (ms * 1e-3), 1)}
            del ha, hb, hc
            lap('pcie')
            # launch-bound regime: a gate on 4096 elements, eager vs captured in a HIP graph.  (Run over the
            # 40-bit prime 2^40-87 so that its kernel instantiations do not mix into the rocprof averages of
            # the headline GF(2^61-1) kernels.)
            from mpyc_amd.engine import CapturedLaunches
            P40 = 2**40 - 87
            ctx40 = FieldContext(P40, device=local_rank)
            small = StepData(ctx40, 4096, t, m, gen)
            lam40 = lagrange(P40, range(1, k + 1))
            small.rec = ctx40.recombine_plan([small.shares.row(j) for j in range(k)], lam40, small.y)

            def small_gate():
                ctx40.split(small.a, small.coef, t, m, out=small.shares, mul_by=small.b)
                small.rec()
            ms_eager = time_launches(lambda s: small_gate(), [0], 200)
            cg = CapturedLaunches(small_gate)
            ms_graph = time_launches(lambda s: cg.replay(), [0], 200)
            kern['gate_p40_n4096_eager_vs_graph'] = {'ms_per_launch': round(ms_eager, 5), 'ms_per_replay': round(ms_graph, 5),
                                                     'achieved': None, 'frac': None, 'unit': 'us',
                                                     'units_per_s': round(4096 / (ms_graph * 1e-3), 1)}
            lap('graph_small_gate')
            # dense product over GF(2^61-1) (finfields.py:1126-1135; the author's np_bnnmnist bottleneck)
            for dim in (2048, 4096):
                from mpyc_amd.engine import DevArray
                Am = DevArray(ctx, uniform_field(gen, dim * dim, P61, ctx.torch_device), dim * dim)
                Bm = DevArray(ctx, uniform_field(gen, dim * dim, P61, ctx.torch_device), dim * dim)
                Cm = ctx.empty(dim * dim)
                ms = time_launches(lambda s: ctx.matmul(Am, Bm, dim, dim, dim, out=Cm), [0], 2)
                macs = float(dim) ** 3
                kern[f'matmul_p61_{dim}'] = dict(roof_mfma(macs, 8, ms), kernel='k_limb_gemm<PM64,8> (8 signed base-256 digits)')
                del Am, Bm, Cm
            # the shape the reference's author names as the np_bnnmnist bottleneck (demos/np_bnnmnist.py:10-15): a
            # 1 x 4096 activation row times a 4096 x 4096 weight matrix (and batches of 4 and 8 rows: column sums of partial
            # products, k_vecmat_partial_col), and the transposed form: matrix x vector, matrix x 4 / 8 columns (k_matvec_sub_col)
            for (mm_, kk_, nn_) in ((1, 4096, 4096), (4, 4096, 4096), (8, 4096, 4096), (4096, 4096, 1), (16384, 4096, 1), (16384, 4096, 4), (16384, 4096, 8), (16, 4096, 4096), (64, 4096, 4096)):
                big = [DevArray(ctx, uniform_field(gen, max(mm_, 4096) * 4096, P61, ctx.torch_device), max(mm_, 4096) * 4096) for _ in range(3)]
                small = DevArray(ctx, uniform_field(gen, 64 * 4096, P61, ctx.torch_device), 64 * 4096)
                outm = ctx.empty(mm_ * nn_)
                if mm_ <= 64 and nn_ == 4096:
                    a_ = DevArray(ctx, small.t[:mm_ * kk_], mm_ * kk_)
                    ms = time_launches(lambda w_: ctx.matmul(a_, w_, mm_, kk_, nn_, out=outm), big, 3)
                else:
                    b_ = DevArray(ctx, small.t[:kk_ * nn_], kk_ * nn_)
                    ms = time_launches(lambda w_: ctx.matmul(w_, b_, mm_, kk_, nn_, out=outm), big, 3)
                byts = eb * (mm_ * kk_ + kk_ * nn_ + mm_ * nn_)
                kern[f'matmul_p61_{mm_}x{kk_}x{nn_}'] = dict(roof(byts, ms), units_per_s=round(mm_ * kk_ * nn_ / (ms * 1e-3), 1),
                                                            algorithmic_bytes_per_unit=None)
                if mm_ in (16, 64):             # matrix cores, rows padded to one 64-row tile
                    kern[f'matmul_p61_{mm_}x{kk_}x{nn_}']['mfma'] = roof_mfma(float(mm_) * kk_ * nn_, 8, ms)
                del big, small, outm
            # dense products over two-limb primes (12 / 16 signed digits, diagonals in passes on the matrix cores)
            for nm_, pw_ in (('p96', 2**96 - 17), ('p128', 2**128 - 173)):
                cw = FieldContext(pw_, device=local_rank)
                dim = 2048
                dtw = torch.int32 if cw.elem_bytes == 12 else torch.int64
                hiw = 2**31 - 1 if cw.elem_bytes == 12 else 2**62
                Aw = DevArray(cw, torch.randint(0, hiw, (dim * dim, cw.limbs), dtype=dtw, device=ctx.torch_device, generator=gen), dim * dim)
                Bw = DevArray(cw, torch.randint(0, hiw, (dim * dim, cw.limbs), dtype=dtw, device=ctx.torch_device, generator=gen), dim * dim)
                cw.reduce(Aw, out=Aw)
                cw.reduce(Bw, out=Bw)
                Cw = cw.empty(dim * dim)
                ms = time_launches(lambda s_: cw.matmul(Aw, Bw, dim, dim, dim, out=Cw), [0], 2)
                kern[f'matmul_{nm_}_{dim}'] = dict(roof_mfma(float(dim) ** 3, 12 if cw.elem_bytes == 12 else 16, ms),
                                                   kernel='k_limb_gemm_wide (12 / 16 signed digits, diagonals in passes)')
                del Aw, Bw, Cw
            lap('matmul')
            # configs[2]: P64, m=7, t=3 (share + recombine from t+1 and 2t+1 rows)
            del sets[1:]
            torch.cuda.empty_cache()
            ctx64 = FieldContext(P64, device=local_rank)
            t2, m2 = 3, 7
            sets64 = [StepData(ctx64, n, t2, m2, gen) for _ in range(3)]
            ms = time_launches(lambda s: ctx64.mul(s.a, s.b, out=s.c), sets64, reps)
            kern['mul_p64'] = dict(roof(3 * eb * n, ms), algorithmic_bytes_per_unit=3 * eb,
                                   units_per_s=round(n / (ms * 1e-3), 1))
            ms = time_launches(lambda s: ctx64.split(s.a, s.coef, t2, m2, out=s.shares), sets64, reps)
            bpu = (1 + t2 + m2) * eb
            kern['split_p64_m7t3'] = dict(roof(bpu * n, ms), algorithmic_bytes_per_unit=bpu,
                                          units_per_s=round(n / (ms * 1e-3), 1))
            # production mode: coefficients from the on-device CSPRNG (never in HBM): 8 read + 56 written
            key = bytes(range(32))
            for rounds in (20, 12, 8):
                ms = time_launches(lambda s: ctx64.split_rng(s.a, t2, m2, key=key, nonce=7, rounds=rounds, out=s.shares),
                                   sets64, reps)
                bpu = (1 + m2) * eb
                kern[f'split_rng_p64_m7t3_chacha{rounds}'] = dict(roof(bpu * n, ms), algorithmic_bytes_per_unit=bpu,
                                                                  units_per_s=round(n / (ms * 1e-3), 1))
            plans64 = {}
            for kk in (t2 + 1, 2 * t2 + 1):
                lam64 = lagrange(P64, range(1, kk + 1))
                plans64[kk] = [ctx64.recombine_plan([s.shares.row(j) for j in range(kk)], lam64, s.y) for s in sets64]
                ms = time_launches(lambda pl: pl(), plans64[kk], reps)
                bpu = (kk + 1) * eb
                kern[f'recombine_p64_k{kk}'] = dict(roof(bpu * n, ms), algorithmic_bytes_per_unit=bpu,
                                                    units_per_s=round(n / (ms * 1e-3), 1))
            # second headline: configs[2] as ONE timed pass per secret -- share generation (m=7,t=3, coefficients
            # supplied: the parity mode) followed by recombination from t+1 = 4 rows (and from 2t+1 = 7 rows)
            cfg2 = {'metric': 'secrets/sec (Shamir share m=7,t=3 + Lagrange recombine) on 10^7 secrets, 64-bit prime',
                    'config': {'workload': 'configs[2]: GF(2^64-189), n = 10^7 secrets, np_random_split(m=7,t=3, coefficients '
                                           'supplied) then np_recombine from k rows', 'n': n, 'prime': '2^64-189', 'm': m2, 't': t2},
                    'unit': 'secrets/s', 'dtype': 'u64'}
            for kk in (t2 + 1, 2 * t2 + 1):
                pairs = list(zip(sets64, plans64[kk]))

                def share_and_recombine(sp):
                    ctx64.split(sp[0].a, sp[0].coef, t2, m2, out=sp[0].shares)
                    sp[1]()
                ms = time_launches(share_and_recombine, pairs, reps)
                bpu = (1 + t2 + m2) * eb + (kk + 1) * eb
                cfg2[f'k{kk}'] = dict(roof(bpu * n, ms), value=round(n / (ms * 1e-3), 1), algorithmic_bytes_per_unit=bpu,
                                      ms_per_pass=round(ms, 5))
                if not torch.equal(sets64[0].y.t, sets64[0].a.t):
                    raise SystemExit('bench parity check failed: recombine(split(s)) != s for configs[2]')
            cfg2['value'] = cfg2[f'k{t2 + 1}']['value']
            cfg2['roofline'] = dict({q: kern['split_p64_m7t3'][q] for q in ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic')},
                                    kernel='k_split<PM64<true,false>,3> (dominant: 88 of the 128 B per secret)',
                                    ms_per_launch=kern['split_p64_m7t3']['ms_per_launch'])
            out['configs2'] = cfg2
            del plans64
            lap('configs2')
            # generic (non pseudo-Mersenne) primes: reciprocal / Montgomery reductions, u32 storage
            from mpyc_amd.engine import DevArray as _DA
            for label, modulus in (('rc64_generic63', 6616326157076047771), ('rc32_p31', 2**31 - 1),
                                   ('mont128_generic', 258797994007609146293811961253269568351)):
                cg_ = FieldContext(modulus, device=local_rank)
                ebg = cg_.elem_bytes
                bufs = []
                for _ in range(3):
                    if ebg == 16:
                        x = torch.randint(0, 2**62, (3, n, 2), dtype=torch.int64, device=ctx.torch_device, generator=gen)
                    elif ebg == 8:
                        x = torch.randint(0, 2**62, (3, n), dtype=torch.int64, device=ctx.torch_device, generator=gen)
                    else:
                        x = torch.randint(0, 2**31 - 1, (3, n), dtype=torch.int32, device=ctx.torch_device, generator=gen)
                    bufs.append(tuple(_DA(cg_, x[i], n) for i in range(3)))
                ms = time_launches(lambda s: cg_.mul(s[0], s[1], out=s[2]), bufs, reps)
                kern[f'mul_{label}'] = dict(roof(3 * ebg * n, ms), algorithmic_bytes_per_unit=3 * ebg,
                                            units_per_s=round(n / (ms * 1e-3), 1))
                del bufs
                torch.cuda.empty_cache()
            lap('generic_primes')
            # configs[4] names GF(2^128) next to GF(2^8): wide binary fields (carry-less products on the integer
            # multiplier; gfpx.py:988-1045), mul / share generation / recombination at the m=7,t=3 setting
            for label, modulus, tail, ebg in (('gf2_64', (1 << 64) | 0x1b, (), 8), ('gf2_128', (1 << 128) | 0x87, (2,), 16)):
                cb_ = FieldContext(modulus, binary=True, device=local_rank)
                bufs = []
                for _ in range(3):
                    x = torch.randint(-2**63, 2**63 - 1, (3, n) + tail, dtype=torch.int64, device=ctx.torch_device, generator=gen)
                    bufs.append(tuple(_DA(cb_, x[i], n) for i in range(3)))
                ms = time_launches(lambda s: cb_.mul(s[0], s[1], out=s[2]), bufs, reps)
                kern[f'mul_{label}'] = dict(roof(3 * ebg * n, ms), algorithmic_bytes_per_unit=3 * ebg, bound_note='integer ALU (carry-less product)',
                                            units_per_s=round(n / (ms * 1e-3), 1))
                if label == 'gf2_128':
                    # no carry-less multiply on gfx950: 9 x 16 v_mad_u64_u32 on "every 4th bit" classes + logic
                    kern['mul_gf2_128']['valu_mads_per_unit'] = 144.0
                cfb = cb_.empty_matrix(t2, n)
                for j in range(t2):
                    cfb.row(j).t.copy_(bufs[j][1].t)
                shb = [cb_.empty_matrix(m2, n) for _ in range(2)]
                ms = time_launches(lambda i_: cb_.split(bufs[i_][0], cfb, t2, m2, out=shb[i_ % 2]), [0, 1, 2], reps)
                bpu = (1 + t2 + m2) * ebg
                kern[f'split_{label}_m7t3'] = dict(roof(bpu * n, ms), algorithmic_bytes_per_unit=bpu, units_per_s=round(n / (ms * 1e-3), 1))
                Fb = gff.GF(ggx.GFpX(2)(modulus))
                for kk in (t2 + 1, 2 * t2 + 1):
                    lamb = [int(v) for v in gth._recombination_vector(Fb, tuple(range(1, kk + 1)), 0)]
                    plans = [cb_.recombine_plan([shb[i_].row(j) for j in range(kk)], lamb, bufs[i_][2]) for i_ in range(2)]
                    ms = time_launches(lambda pl: pl(), plans, reps)
                    bpu = (kk + 1) * ebg
                    kern[f'recombine_{label}_k{kk}'] = dict(roof(bpu * n, ms), algorithmic_bytes_per_unit=bpu,
                                                            bound_note='Lagrange coefficients of parties 1..k at 0 (all ones for k = 7: '
                                                                       'plain XOR of the rows); LDS nibble tables for dense ones',
                                                            coefficients=[hex(v) for v in lamb][:4], units_per_s=round(n / (ms * 1e-3), 1))
                    # worst case for the same shape: k dense (random) coefficients -> every row goes through the LDS tables
                    import random as _rnd
                    rg_ = _rnd.Random(1000 + kk)                      # ONE generator: k DISTINCT coefficients = k tables (until round 6 a
                    #                                                   fresh generator per draw made them all equal: one or two tables)
                    lamd = [rg_.randrange(2, 1 << (8 * ebg)) for _ in range(kk)]
                    plans = [cb_.recombine_plan([shb[i_].row(j) for j in range(kk)], lamd, bufs[i_][2]) for i_ in range(2)]
                    ms = time_launches(lambda pl: pl(), plans, reps)
                    kern[f'recombine_{label}_k{kk}_dense'] = dict(roof(bpu * n, ms), algorithmic_bytes_per_unit=bpu,
                                                                  bound_note='LDS tables (16 B x 32 look-ups per row) + VALU',
                                                                  units_per_s=round(n / (ms * 1e-3), 1))
                cb_.split(bufs[0][0], cfb, t2, m2, out=shb[0])
                lamb = [int(v) for v in gth._recombination_vector(Fb, tuple(range(1, t2 + 2)), 0)]
                if not torch.equal(cb_.recombine([shb[0].row(j) for j in range(t2 + 1)], lamb).t, bufs[0][0].t):
                    raise SystemExit(f'bench parity check failed for {label} split/recombine')
                del bufs, cfb, shb, plans
                torch.cuda.empty_cache()
            # GF(2^32) (four-byte storage since round 6, fold reduction): element-wise product
            c32_ = FieldContext((1 << 32) | 0x8d, binary=True, device=local_rank)
            bufs = []
            for _ in range(3):
                x = torch.randint(-2**31, 2**31 - 1, (3, n), dtype=torch.int32, device=ctx.torch_device, generator=gen)
                bufs.append(tuple(_DA(c32_, x[i], n) for i in range(3)))
            ms = time_launches(lambda s: c32_.mul(s[0], s[1], out=s[2]), bufs, reps)
            kern['mul_gf2_32'] = dict(roof(3 * 4 * n, ms), algorithmic_bytes_per_unit=12, units_per_s=round(n / (ms * 1e-3), 1))
            del bufs
            torch.cuda.empty_cache()
            lap('gf2w')
            # configs[3] shape on ONE GPU: 128-bit prime (two limbs), gate = mul + split(m=7,t=3) + recombine(k=7)
            del sets64[:]
            torch.cuda.empty_cache()
            P128 = 2**128 - 173
            ctx128 = FieldContext(P128, device=local_rank)

            def u128(rows):
                return u128_rows(rows, n, ctx.torch_device, gen)

            class Set128:
                def __init__(s):
                    from mpyc_amd.engine import DevArray
                    ab = u128(2)
                    s.a, s.b = DevArray(ctx128, ab[0], n), DevArray(ctx128, ab[1], n)
                    s.c = ctx128.empty(n)
                    s.coef = ctx128.empty_matrix(t2, n)
                    cf_ = u128(t2)
                    for j in range(t2):
                        s.coef.row(j).t.copy_(cf_[j])
                    s.shares = ctx128.empty_matrix(m2, n)
                    s.y = ctx128.empty(n)
                    s.rec = None
            sets128 = [Set128() for _ in range(2)]
            eb2 = 16
            ms = time_launches(lambda s: ctx128.mul(s.a, s.b, out=s.c), sets128, reps)
            kern['mul_p128'] = dict(roof(3 * eb2 * n, ms), algorithmic_bytes_per_unit=3 * eb2,
                                    units_per_s=round(n / (ms * 1e-3), 1))
            ms = time_launches(lambda s: ctx128.split(s.c, s.coef, t2, m2, out=s.shares), sets128, reps)
            bpu = (1 + t2 + m2) * eb2
            kern['split_p128_m7t3'] = dict(roof(bpu * n, ms), algorithmic_bytes_per_unit=bpu,
                                           units_per_s=round(n / (ms * 1e-3), 1))
            lam128 = lagrange(P128, range(1, 2 * t2 + 2))
            for s in sets128:
                s.rec = ctx128.recombine_plan([s.shares.row(j) for j in range(2 * t2 + 1)], lam128, s.y)
            ms = time_launches(lambda s: s.rec(), sets128, reps)
            bpu = (2 * t2 + 2) * eb2
            kern['recombine_p128_k7'] = dict(roof(bpu * n, ms), algorithmic_bytes_per_unit=bpu,
                                             units_per_s=round(n / (ms * 1e-3), 1))
            if not torch.equal(sets128[0].y.t, sets128[0].c.t):
                raise SystemExit('bench parity check failed for P128 gate')
            gate_ms = sum(kern[q]['ms_per_launch'] for q in ('mul_p128', 'split_p128_m7t3', 'recombine_p128_k7'))
            kern['gate_p128_m7t3'] = {'gates_per_s': round(n / (gate_ms * 1e-3), 1), 'algorithmic_bytes_per_unit': 352,
                                      'achieved': round(352 * n / (gate_ms * 1e-3) / 1e9, 1), 'unit': 'GB/s',
                                      'frac': round(352 * n / (gate_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                      'ms_per_launch': round(gate_ms, 5)}
            del sets128[:]
            torch.cuda.empty_cache()
            lap('configs3_p128')
            # three-limb primes (SecInt(97..160) defaults; np_lpsolver's largest dataset runs over the 136-bit one):
            # 24-byte elements, one per lane, the same kernels instantiated on the PM192 policy
            from mpyc_amd.finfields import find_prime_root
            from mpyc_amd.engine import DevArray
            P136 = find_prime_root(136)[0]
            ctx136 = FieldContext(P136, device=local_rank)
            eb3 = 24

            def rows136():
                a_ = DevArray(ctx136, torch.randint(0, 2**62, (n, 3), dtype=torch.int64, device=ctx.torch_device, generator=gen), n)
                return ctx136.reduce(a_, out=a_)
            a3, b3, c3 = rows136(), rows136(), ctx136.empty(n)
            sh3 = ctx136.empty_matrix(m, n)
            ms = time_launches(lambda s_: ctx136.mul(a3, b3, out=c3), [0], reps)
            kern['mul_p136'] = dict(roof(3 * eb3 * n, ms), algorithmic_bytes_per_unit=3 * eb3, units_per_s=round(n / (ms * 1e-3), 1))
            ms = time_launches(lambda s_: ctx136.split_rng(c3, t, m, key=bytes(range(32)), nonce=5, out=sh3), [0], reps)
            bpu = (1 + m) * eb3
            kern['split_rng_p136_m3t1_chacha20'] = dict(roof(bpu * n, ms), algorithmic_bytes_per_unit=bpu,
                                                        units_per_s=round(n / (ms * 1e-3), 1))
            y3 = ctx136.empty(n)
            rec3 = ctx136.recombine_plan([sh3.row(j) for j in range(k)], lagrange(P136, range(1, k + 1)), y3)
            ms = time_launches(lambda s_: rec3(), [0], reps)
            bpu = (k + 1) * eb3
            kern['recombine_p136_k3'] = dict(roof(bpu * n, ms), algorithmic_bytes_per_unit=bpu, units_per_s=round(n / (ms * 1e-3), 1))
            if not torch.equal(y3.t, c3.t):
                raise SystemExit('bench parity check failed for the 136-bit field')
            del a3, b3, c3, sh3, y3, rec3
            torch.cuda.empty_cache()
            lap('p136')
            # the 80-bit prime of the default SecFxp() (12-byte storage): product, recombination from 2t+1 rows, and the
            # inverse square root that np_random_bits takes of every opened square (runtime.py:4243-4273: f x n of them per
            # fixed-point product of n elements -- two thirds of that product's GPU time until round 6's digit chains)
            P80 = find_prime_root(80)[0]
            ctx80 = FieldContext(P80, device=local_rank)
            eb80 = ctx80.elem_bytes                                       # 12

            def rows80():
                a_ = DevArray(ctx80, torch.randint(0, 2**31 - 1, (n, 3), dtype=torch.int32, device=ctx.torch_device, generator=gen), n)
                return ctx80.reduce(a_, out=a_)
            a8_, b8_, c8_ = rows80(), rows80(), ctx80.empty(n)
            ms = time_launches(lambda s_: ctx80.mul(a8_, b8_, out=c8_), [0], reps)
            kern['mul_p80'] = dict(roof(3 * eb80 * n, ms), algorithmic_bytes_per_unit=3 * eb80, units_per_s=round(n / (ms * 1e-3), 1))
            sh80 = ctx80.empty_matrix(m2, n)
            ctx80.split_rng(c8_, t2, m2, key=bytes(range(32)), nonce=9, out=sh80)
            for kk in (t2 + 1, 2 * t2 + 1):
                y80 = ctx80.empty(n)
                rec80 = ctx80.recombine_plan([sh80.row(j) for j in range(kk)], lagrange(P80, range(1, kk + 1)), y80)
                ms = time_launches(lambda s_: rec80(), [0], reps)
                bpu = (kk + 1) * eb80
                kern[f'recombine_p80_k{kk}'] = dict(roof(bpu * n, ms), algorithmic_bytes_per_unit=bpu, units_per_s=round(n / (ms * 1e-3), 1))
                if not torch.equal(y80.t, c8_.t):
                    raise SystemExit('bench parity check failed for the 80-bit field')
            e_inv_sqrt = (3 * P80 - 5) >> 2                               # a^((3p-5)/4) = a^(-1/2) for p = 3 mod 4 (finfields.py:1424-1437)
            ms = time_launches(lambda s_: ctx80.pow(a8_, e_inv_sqrt, out=c8_), [0], max(2, reps // 4))
            kern['inv_sqrt_p80'] = dict(roof(2 * eb80 * n, ms), algorithmic_bytes_per_unit=2 * eb80, units_per_s=round(n / (ms * 1e-3), 1),
                                        bound='valu', bound_note='~106 products per element, in three 27-bit digits (fields.hpp DigitChain)')
            ms = time_launches(lambda s_: ctx80.inv(a8_, out=c8_, check_zero=False), [0], reps)
            kern['inv_p80'] = dict(roof(2 * eb80 * n, ms), algorithmic_bytes_per_unit=2 * eb80, units_per_s=round(n / (ms * 1e-3), 1),
                                   bound='valu', bound_note='batched inverse in digits: 32 elements per thread share one exponentiation (k_inv_digits)')
            del a8_, b8_, c8_, sh80, y80, rec80
            torch.cuda.empty_cache()
            lap('p80')
            # configs[4]: GF(2^8) (AES field): element-wise mul and the local S-box layer
            ctx8 = FieldContext(0x11b, binary=True, device=local_rank)
            # demos/np_aes.py:23-33: A = circulant([1,0,0,0,1,1,1,1]) (row j = first row rolled by j), B = 0x63
            r_ = [1, 0, 0, 0, 1, 1, 1, 1]
            rows8 = [sum(r_[(c_ - j_) % 8] << c_ for c_ in range(8)) for j_ in range(8)]
            b8 = 0x63
            for n8 in (1_000_000, 1_000_000_000):
                from mpyc_amd.engine import DevArray
                bufs = []
                for _ in range(3 if n8 > 10**8 else 8):
                    x = torch.randint(0, 256, (3, n8), dtype=torch.uint8, device=ctx.torch_device, generator=gen)
                    bufs.append((DevArray(ctx8, x[0], n8), DevArray(ctx8, x[1], n8), DevArray(ctx8, x[2], n8)))
                tag = '1e6' if n8 == 1_000_000 else '1e9'
                ms = time_launches(lambda s: ctx8.mul(s[0], s[1], out=s[2]), bufs, reps if n8 < 10**8 else 3)
                kern[f'gf256_mul_{tag}'] = dict(roof(3 * n8, ms), algorithmic_bytes_per_unit=3,
                                                units_per_s=round(n8 / (ms * 1e-3), 1))
                ms = time_launches(lambda s: ctx8.sbox(s[0], rows8, b8, out=s[2]), bufs, reps if n8 < 10**8 else 3)
                kern[f'gf256_sbox_{tag}'] = dict(roof(2 * n8, ms), algorithmic_bytes_per_unit=2,
                                                 units_per_s=round(n8 / (ms * 1e-3), 1))
                del bufs
                torch.cuda.empty_cache()
            lap('gf256_public')
            # configs[4] proper: the np_aes S-box layer over SECURE bytes (demos/np_aes.py:37-43) -- x^254 by 11 GRR
            # multiplications, bit decomposition with shared random bits, GF(2) affine map on bit shares,
            # recomposition -- the compute of all m = 3 parties (t = 1) on this GPU, no networking
            # (mpyc_amd/protocols.py; opened result checked against the public S-box kernel).
            F8 = gff.GF(ggx.GFpX(2)(0x11b))
            A8 = [[(rows8[r__] >> c__) & 1 for c__ in range(8)] for r__ in range(8)]
            B8 = [(b8 >> r__) & 1 for r__ in range(8)]
            for n8, tag in ((1_000_000, '1e6'), (100_000_000, '1e8')):
                xpub = DevArray(ctx8, torch.randint(0, 256, (n8,), dtype=torch.uint8, device=ctx.torch_device, generator=gen), n8)
                xs = protocols.as_matrix(ctx8, protocols.share(ctx8, xpub, 1, 3))
                rb = DevArray(ctx8, torch.randint(0, 2, (8 * n8,), dtype=torch.uint8, device=ctx.torch_device, generator=gen), 8 * n8)
                rbits = protocols.as_matrix(ctx8, protocols.share(ctx8, rb, 1, 3))
                del rb
                want8 = ctx8.sbox(xpub, rows8, b8).t

                def opened(mtx):
                    return protocols.open_(ctx8, F8, [mtx.row(i_) for i_ in range(3)], 1).t
                for fused_ in (True, False):
                    res = protocols.sbox_layer_all(ctx8, F8, xs, rbits, 1, A8, B8, fused=fused_)
                    if not torch.equal(opened(res), want8):
                        raise SystemExit(f'bench parity check failed for the secure S-box layer (fused={fused_})')
                    del res
                # (a) ONE kernel for the whole layer (ffgpu_gf256_sbox_layer): the parties' shares of four bytes travel
                # through the 11 gates, the opening and the affine fold in registers.  HBM traffic 10 m = 30 B per secure
                # byte; the kernel is bound by VALU work (ChaCha20 for 33 coefficient words + 33 GF(2^8) products per 4
                # bytes; lane-operations per secure byte measured, profiles/r04_valu.md), so both fractions are given
                ms = time_launches(lambda s_: protocols.sbox_layer_all(ctx8, F8, xs, rbits, 1, A8, B8), [0], 5 if n8 < 10**8 else 2)
                # (VALU-bound: SQ_INSTS_VALU x 64 lanes / n of this kernel at this size -- a fresh, interleaved keystream per step at
                # 10^6, the continued keystream at 10^8 -- and `valu_frac` come from annotate_valu)
                kern[f'secure_sbox_layer_m3t1_{tag}'] = dict(
                    roof(30 * n8, ms), algorithmic_bytes_per_unit=30, units_per_s=round(n8 / (ms * 1e-3), 1), kernels_per_layer=1,
                    bound='valu', hbm_frac_at_unfused_269B=round(269 * n8 / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                    note='frac = HBM fraction for the 30 B the fused kernel moves; hbm_frac_at_unfused_269B = the same time priced at the '
                         '269 B per secure byte of the 13-kernel composition (the accounting of rounds 1-2)')
                # (b) the 13-launch composition (11 batched chain gates + masked opening + bits/affine/fold), 269 B per secure
                # byte: (4+6+6+7+6+6+7+6+9+9+6) x 3 = 216 for the gates, 23 for the opening, 30 for the fold
                bpu = 269
                ms = time_launches(lambda s_: protocols.sbox_layer_all(ctx8, F8, xs, rbits, 1, A8, B8, fused=False), [0], 5 if n8 < 10**8 else 2)
                kern[f'secure_sbox_layer_m3t1_{tag}_13_kernels'] = dict(roof(bpu * n8, ms), algorithmic_bytes_per_unit=bpu,
                                                                        units_per_s=round(n8 / (ms * 1e-3), 1), kernels_per_layer=13)
                if n8 <= 10**6:
                    # the same layers captured once in a HIP graph (device-resident generator state: fresh
                    # randomness on every replay) -- the launch-bound regime is where graphs pay
                    for fused_, key_ in ((True, f'secure_sbox_layer_m3t1_{tag}_hipgraph'), (False, f'secure_sbox_layer_m3t1_{tag}_13_kernels_hipgraph')):
                        st8 = ctx8.rng_state()
                        cg8 = CapturedLaunches(lambda: protocols.sbox_layer_all(ctx8, F8, xs, rbits, 1, A8, B8, rng=st8, fused=fused_))
                        cg8.replay()
                        if not torch.equal(opened(cg8.result), want8):
                            raise SystemExit('bench parity check failed for the graph-replayed secure S-box layer')
                        msg = time_launches(lambda s_: cg8.replay(), [0], 20)
                        bb = 30 if fused_ else 269
                        kern[key_] = dict(roof(bb * n8, msg), algorithmic_bytes_per_unit=bb, units_per_s=round(n8 / (msg * 1e-3), 1))
                        if fused_:
                            kern[key_].update(bound='valu', valu_counts_of=f'secure_sbox_layer_m3t1_{tag}',
                                              hbm_frac_at_unfused_269B=round(269 * n8 / (msg * 1e-3) / 1e9 / HBM_PEAK_GBS, 4))
                        del cg8
                    # one launch per party and step (what each MPyC party does in its own process): 49 launches, 345 B
                    xl, rl = [xs.row(i_) for i_ in range(3)], [rbits.row(i_) for i_ in range(3)]
                    ms = time_launches(lambda s_: protocols.sbox_layer(ctx8, F8, xl, rl, 1, A8, B8), [0], 5)
                    kern[f'secure_sbox_layer_m3t1_{tag}_launch_per_party'] = dict(roof(345 * n8, ms), algorithmic_bytes_per_unit=345,
                                                                                   units_per_s=round(n8 / (ms * 1e-3), 1),
                                                                                   kernels_per_layer=49)
                del xs, rbits, xpub, want8
                torch.cuda.empty_cache()
            lap('secure_sbox_layer')
            # AES-128 encryption of secret-shared blocks under secret-shared round keys (np_aes.py:75-86): 10 S-box
            # layers + ShiftRows/MixColumns (row relabelling + the small-matrix kernel) + AddRoundKey, all 3 parties
            for nblk in (62_500, 6_250_000):
                nb = 16 * nblk
                kpub = DevArray(ctx8, torch.randint(0, 256, (nb,), dtype=torch.uint8, device=ctx.torch_device, generator=gen), nb)
                ppub = DevArray(ctx8, torch.randint(0, 256, (nb,), dtype=torch.uint8, device=ctx.torch_device, generator=gen), nb)
                Ks = [protocols.share(ctx8, kpub, 1, 3) for _ in range(11)]          # any 11 shared round keys
                ps = protocols.share(ctx8, ppub, 1, 3)
                pools = []
                for _ in range(10):
                    rb = DevArray(ctx8, torch.randint(0, 2, (8 * nb,), dtype=torch.uint8, device=ctx.torch_device, generator=gen), 8 * nb)
                    pools.append(protocols.share(ctx8, rb, 1, 3))
                    del rb

                def run_aes():
                    it = iter(pools)
                    return protocols.aes128_encrypt(ctx8, F8, Ks, ps, nblk, lambda nbytes: next(it), 1, A8, B8)
                ms = time_launches(lambda s_: run_aes(), [0], 2)
                kern[f'secure_aes128_encrypt_m3t1_{nblk}_blocks'] = {'ms_per_launch': round(ms, 4), 'unit': 'blocks/s', 'bound': 'hbm/alu',
                                                                       'achieved': round(nblk / (ms * 1e-3), 1), 'frac': None,
                                                                       'units_per_s': round(nblk / (ms * 1e-3), 1)}
                del Ks, ps, pools, kpub, ppub
                torch.cuda.empty_cache()
            lap('secure_aes')
        # dominant kernel of the timed step = the one with the largest share of step time
        step_kernels = ['mul_split_fused_p61_m3t1', 'recombine_p61_k3']
        dom = max(step_kernels, key=lambda q: kern[q]['ms_per_launch'])
        out['roofline'] = dict({kk_: vv for kk_, vv in kern[dom].items()
                                if kk_ in ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic')},
                               kernel=kern[dom].get('kernel'), name=dom,
                               ms_per_launch=kern[dom]['ms_per_launch'],
                               frac_of_measured_copy=round(kern[dom]['achieved'] / kern['device_copy']['achieved'], 4))
        # HBM traffic per launch from PMC counters (collected separately with rocprofv3 --pmc, see
        # profiles/r03_pmc_traffic.md for the command, units and the gfx950 FETCH_SIZE correction)
        def annotate_traffic():
            try:
                pmc_file = next(f_ for f_ in ('r06_pmc_traffic.json', 'r05_pmc_traffic.json', 'r04_pmc_traffic.json', 'r03_pmc_traffic.json', 'r02_pmc_traffic.json', 'r01_pmc_traffic.json')
                                if os.path.exists(os.path.join(ROOT, 'profiles', f_)))
                with open(os.path.join(ROOT, 'profiles', pmc_file)) as fh:
                    pmc = json.load(fh)
                names = {'mul_split_fused_p61_m3t1': 'k_split<PM64<false, true>, 1, true, true, false, false>',
                         'mul_p61': 'k_ew2<PM64<false, true>, 2, true>',
                         'split_p61_m3t1': 'k_split<PM64<false, true>, 1, false, true, false, false>',
                         'recombine_p61_k3': 'k_recombine<PM64<false, true>, 3, true>',
                         'mul_p64': 'k_ew2<PM64<true, false>, 2, true>',
                         'split_p64_m7t3': 'k_split<PM64<true, false>, 3, false, true, false, false>',
                         'recombine_p64_k7': 'k_recombine<PM64<true, false>, 7, true>',
                         'mul_p128': 'k_ew2<PM128<true>, 2, true>',
                         'split_p128_m7t3': 'k_split<PM128<true>, 3, false, true, false, false>',
                         'recombine_p128_k7': 'k_recombine<PM128<true>, 7, true>',
                         'device_copy': 'k_copy16'}
                for q, kn in names.items():
                    if q in kern and kn in pmc and n == 10_000_000:
                        kern[q]['traffic'] = pmc[kn]['traffic_bytes']
                if n == 10_000_000 and names.get(dom) in pmc:
                    out['roofline']['traffic'] = pmc[names[dom]]['traffic_bytes']
                    out['roofline']['traffic_source'] = f'profiles/{pmc_file[:-5]}.md (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE)'
            except (OSError, ValueError, StopIteration):
                pass
        annotate_traffic()
        out['roofline']['bytes_per_launch'] = kern[dom]['bytes_per_launch']
        out['kernels'] = kern
        out['mulmod_per_s_1gpu'] = kern['mul_p61']['units_per_s']

    def after_the_collectives():
        """Rank 0 alone, N = 1 only (the contract's `cpu_baseline` rule; at N > 1 the other ranks have left and the driver's
        N = 1 run already holds these sections): every measurement beyond the step's own kernels, the reference CPU path
        and the API-level section.  A failure here never costs the headline line."""
        if rank != 0 or world != 1:
            return
        if args.full:
            LEG_BUDGET.update(deadline=None, leg_timeout=300.0)
        else:
            LEG_BUDGET.update(deadline=time.perf_counter() + args.budget, leg_timeout=60.0)
        if not args.no_extras:
            try:
                lap('headline_and_multi_gpu_leg')
                out['valu_peak'] = measure_valu_peak(ctx)
                lap('valu_peak')
                optional_measurements()
                annotate_traffic()
                annotate_valu(kern, out)
            except Exception as exc:          # noqa: BLE001 -- report, keep the main result
                out['extras_error'] = f'{type(exc).__name__}: {exc}'
            if 'configs2' in out and 'split_p64_m7t3' in kern:
                out['configs2']['roofline']['traffic'] = kern['split_p64_m7t3'].get('traffic')
        if not args.no_cpu_baseline:
            try:
                lap('misc')
                out['cpu_baseline'] = cpu_baseline(n, t, m, lam)
                lap('cpu_baseline')
            except Exception as exc:          # noqa: BLE001
                out['cpu_baseline'] = {'error': f'{type(exc).__name__}: {exc}'}
        if not args.no_api_leg and not args.no_extras:
            torch.cuda.empty_cache()
            try:
                out['configs0'] = list_path_leg()
                lap('configs0_list_path')
            except Exception as exc:          # noqa: BLE001 -- report, keep the main result
                out['configs0'] = {'error': f'{type(exc).__name__}: {exc}'}
            try:
                out['api'] = api_leg(n)
                lap('api_leg')
            except Exception as exc:          # noqa: BLE001 -- report, keep the main result
                out['api'] = {'error': f'{type(exc).__name__}: {exc}'}

    # configs[3] on all N ranks: element-sharded P128 gate and the party-major exchange (the one collective).
    # It runs LAST and under a watchdog: whatever happens in the collectives (a rank failing, a transport that
    # hangs), rank 0 still prints the line with everything measured above.
    def finish(code=None):
        if rank == 0:
            emit(out)
        if code is not None:
            sys.stdout.flush()
            os._exit(code)

    import threading
    collectives_done = threading.Event()
    # N > 1: a leg that has never met this node's transport must not cost minutes of the scaling run -- 60 s, then the line
    # goes out with what was measured (headline + roofline are complete before the first collective of the leg)
    limit = float(os.environ.get('FFGPU_BENCH_LEG_TIMEOUT', '60' if world > 1 else '300'))

    def watchdog():
        if not collectives_done.wait(limit):
            out.setdefault('multi_gpu', {'error': f'multi-GPU section did not finish within {limit:.0f} s (rank {rank})'})
            finish(0)
    threading.Thread(target=watchdog, daemon=True).start()
    if not args.no_multi_gpu_leg:
        full = args.layout == 'party-major'
        try:
            leg = multi_gpu_leg(dist, rank, world, local_rank, backend, n,
                                args.steps if full else max(2, min(args.steps, 10)), args.warmup if full else max(1, min(args.warmup, 2)), lagrange)
        except torch.OutOfMemoryError as exc:
            leg = {'error': f'OutOfMemoryError: {exc}'}
        except Exception as exc:          # noqa: BLE001 -- the other ranks may now be waiting in a collective: end here
            out['multi_gpu'] = {'error': f'{type(exc).__name__}: {exc}'}
            collectives_done.set()
            finish(0)
        out['multi_gpu'] = leg
        if isinstance(leg.get('parity'), dict) and not leg['parity']['passed_on_every_rank']:
            invalid.append('invalid: a parity check of the multi-GPU section failed on a rank')
            out['scaling'] = invalid[-1]
        torch.cuda.empty_cache()

    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    collectives_done.set()
    after_the_collectives()
    # OPT-IN (--parties-on-gpus / FFGPU_BENCH_PARTIES_ON_GPUS=1; N >= 3 GPUs): the m-party protocol itself with ONE PARTY PER
    # GPU -- three MPyC party processes under install() on GPUs 0, 1, 2 (MPYC_AMD_DEVICE=party), share rows crossing between
    # the GPUs as interprocess handles.  It has never run on more than one GPU, so it is not part of a default run; as
    # subprocesses with their own timeouts, after every collective is done; an error is reported in the object, never raised.
    if rank == 0 and world >= 3 and args.parties_on_gpus and not args.no_api_leg:
        torch.cuda.empty_cache()
        try:
            out['api_parties_on_gpus'] = api_leg(n, parties_on_gpus=True)
        except Exception as exc:          # noqa: BLE001
            out['api_parties_on_gpus'] = {'error': f'{type(exc).__name__}: {exc}'}
    finish(3 if invalid else None)


if __name__ == '__main__':
    main()
