"""bench.py's stdout protocol: the LAST line is a compact strict-JSON object that fits the driver's ~8 KB stdout tail
whatever the run measured; everything else travels in the `# detail ` line and bench_detail.json."""
import io
import json
import os
import sys
from contextlib import redirect_stdout

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import bench  # noqa: E402  (imports torch; needs no GPU)


def canned(n_kernels=120, n_ranks=8, fat=400):
    """A result dict at least as large as anything bench.py has produced (r03: 75 kernel rows, 28.8 KB)."""
    row = {'bound': 'hbm', 'achieved': 6183.3, 'peak': 8000.0, 'unit': 'GB/s', 'frac': 0.7729, 'ms_per_launch': 0.07763,
           'bytes_per_launch': 480000000, 'traffic': 480041963, 'kernel': 'k_split<PM64<false,true>,T=1,fused mul,nt>',
           'algorithmic_bytes_per_unit': 48, 'units_per_s': 1.2e11, 'note': 'x' * fat}
    kern = {f'kernel_row_number_{i}_with_a_long_name': dict(row) for i in range(n_kernels)}
    kern['inv_p61'] = dict(row, bound='valu', valu_frac=0.9, frac=0.4)
    kern['recombine_gf2_128_k7_dense'] = dict(row, bound='lds+valu', valu_frac=0.5, frac=0.45)
    kern['prss_share_p61_m7t3'] = {'ms_per_launch': 98.9, 'bound': 'host', 'unit': 'GB/s', 'achieved': 11.3, 'frac': None}
    kern['mul_p61_pcie_inclusive'] = {'ms_per_launch': 4.4, 'bound': 'pcie', 'unit': 'GB/s', 'achieved': 54.0, 'frac': None}
    kern['secure_aes128_encrypt_m3t1_62500_blocks'] = {'ms_per_launch': 3.2, 'bound': 'hbm/alu', 'unit': 'blocks/s', 'frac': None}
    leg = {'n': 10**7, 'parties': 3, 'ms_per_rep': 1.17, 'elements_per_s': 6.6e9, 'note': 'y' * fat}
    return {
        'metric': 'field-ops/sec (modmul + share+recombine) on 10^7-elt SecFld array', 'value': 226585726170.5,
        'unit': 'field-ops/s', 'n_gpus': n_ranks, 'steps': 20, 'warmup': 5, 'ms_per_step': 0.1324, 'higher_is_better': True,
        'scaling': 'weak', 'vs_baseline': None, 'dtype': 'u64', 'data': 'synthetic',
        'config': {'workload': 'configs[1]: ' + 'w' * 900, 'n_per_gpu': 10**7, 'prime': '2^61-1', 'm': 3, 't': 1, 'k': 3,
                   'field_ops_per_step': 3 * 10**7, 'buffer_sets': 4, 'parallelism': 'element-sharded x8, no collective'},
        'distributed': {'backend': 'nccl', 'world_size': n_ranks, 'collective_library': 'RCCL', 'rccl_version': '2.26.6',
                        'distinct_devices': n_ranks,
                        'ranks': [{'rank': r, 'local_rank': r, 'device': f'cuda:{r} (AMD Instinct MI355X)',
                                   'pci_bus_id': f'0000:{r:02x}:00.0'} for r in range(n_ranks)]},
        'unfused': {'value': 1.9e11, 'unit': 'field-ops/s', 'ms_per_step': 0.157, 'note': 'z' * fat},
        'roofline': dict(row, name='mul_split_fused_p61_m3t1', frac_of_measured_copy=1.01, traffic_source='s' * fat),
        'kernels': kern, 'mulmod_per_s_1gpu': 2.55e11,
        'configs2': {'k4': {'value': 1e10, 'frac': 0.7, 'ms_per_pass': 0.21}, 'k7': {'value': 9e9}, 'roofline': dict(row, name='split_p64_m7t3'),
                     'note': 'c' * 3000},
        'configs0': {'workload': 'configs[0]: list path', 'mirror_secrets_per_s': 5e5, 'reference_secrets_per_s': 4e5, 'parity': 'p' * 90},
        'valu_peak': {'lane_ops_per_s': 3.6e13, 'mad_u64_u32_lane_ops_per_s': 1.8e13, 'shader_clock_mhz': 2250.0,
                      'source': 'ffgpu_valu_probe in this run', 'x': 'v' * fat},
        'cpu_baseline': {'value': 28694617.3, 'unit': 'field-ops/s', 'cores': 32, 'procs': 32, 'host_cores': 128,
                         'kind': 'reference', 'sample': 's' * 2000, 'value_1core': 2491105.2, 'port_value': 3.5e8,
                         'port_cores': 128, 'process_count_probe': [{'procs': p, 'field_ops_per_s': 1.0} for p in range(64)]},
        'api': dict({f'leg{i}': dict(leg) for i in range(20)}, m3_1e7_ipc=dict(leg), m1_1e7=dict(leg), elements_per_s=7.4e10,
                    gpu_busy_frac=0.53, gpu_busy_frac_1e8=0.89, vs_reference_m1=11607.2, gpu_busy_frac_m3_1e7_ipc=0.35,
                    note='n' * 3000),
        'multi_gpu': {'config': {'workload': 'configs[3]: ' + 'g' * 500, 'backend': 'nccl'},
                      'gate_sharded': {'ms_per_step': 0.56, 'gates_per_s': 1.78e10, 'frac_of_hbm_peak': 0.71, 'x': 'q' * fat},
                      'party_major_all_to_all': {'ms_per_step': 0.21, 'secrets_per_s': 4.7e10, 'exchange_share': 0.04},
                      'party_major_all_to_all_pipelined': {'ms_per_step': 0.23, 'secrets_per_s': 4.4e10},
                      'party_major_allgather': {'ms_per_step': 0.62, 'secrets_per_s': 1.6e10, 'exchange_share': 0.69}},
        'extras_error': 'e' * 5000,
    }


REQUIRED = ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline',
            'dtype', 'data', 'config', 'roofline', 'cpu_baseline', 'distributed')


def test_compact_line_fits_the_driver_tail_and_parses():
    for kw in ({}, {'n_kernels': 1000, 'fat': 4000}, {'n_kernels': 3, 'n_ranks': 1}):
        out = canned(**kw)
        text = bench.compact_line(out)
        assert '\n' not in text
        assert len(text) < bench.COMPACT_LIMIT, len(text)
        line = json.loads(text)
        for key in REQUIRED:
            assert key in line, key
        assert line['value'] == out['value'] and line['ms_per_step'] == out['ms_per_step']
        assert line['config']['workload'].startswith('configs[1]') and line['config']['n_per_gpu'] == 10**7
        for key in ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic', 'kernel', 'ms_per_launch', 'bytes_per_launch'):
            assert key in line['roofline'], key
        for key in ('value', 'unit', 'cores', 'kind', 'procs', 'host_cores', 'value_1core', 'port_value', 'sample'):
            assert key in line['cpu_baseline'], key
        assert line['distributed']['world_size'] == out['n_gpus'] == len(line['distributed']['ranks'])
        assert line['api']['m3_1e7_ipc']['ms_per_rep'] == 1.17
        assert 'kernels' not in line and 'kernel_fracs' not in line
        # no null where a number was measured ...
        assert line['configs2']['k4_secrets_per_s'] == 1e10 and line['configs2']['k7_secrets_per_s'] == 9e9
        assert line['roofline']['traffic_source']
        assert line['configs0']['mirror_secrets_per_s'] == 5e5 and line['valu_peak']['shader_clock_mhz'] == 2250.0
        # ... and no key that mixes two kinds of fraction (weak 5a): a row is in exactly one of the two maps, rows without a
        # roofline meaning (host / PCIe / whole-protocol rows) in neither, and no 0.0 placeholders
        if kw.get('n_kernels', 120) <= 120:
            hb, va = line['hbm_fracs'], line['valu_fracs']
            assert (len(hb) == 3) if kw.get('n_kernels') == 3 else (0 < len(hb) < 120 and line['hbm_fracs_rows_cut'] == 120 - len(hb))
            assert not set(hb) & set(va)
            assert va == {'inv_p61': 0.9, 'recombine_gf2_128_k7_dense': 0.5}
            assert 'inv_p61' not in hb and all(v == 0.7729 for v in hb.values())
            for name in ('prss_share_p61_m7t3', 'mul_p61_pcie_inclusive', 'secure_aes128_encrypt_m3t1_62500_blocks'):
                assert name not in hb and name not in va


def test_compact_line_without_optional_sections():
    out = canned()
    for key in ('kernels', 'api', 'multi_gpu', 'configs2', 'cpu_baseline', 'roofline', 'unfused', 'extras_error'):
        out.pop(key)
    line = json.loads(bench.compact_line(out))
    assert line['metric'] == out['metric'] and 'roofline' not in line


def test_emit_prints_detail_first_and_the_compact_line_last(tmp_path):
    out = canned()
    buf = io.StringIO()
    with redirect_stdout(buf):
        bench.emit(out, detail_path=str(tmp_path / 'bench_detail.json'))
    lines = buf.getvalue().splitlines()
    assert len(lines) == 2
    assert lines[0].startswith('# detail {')            # not a bare JSON line: a line-by-line parser skips it
    detail = json.loads(lines[0][len('# detail '):])
    assert len(detail['kernels']) == len(out['kernels'])
    last = json.loads(lines[-1])
    assert len(lines[-1]) < bench.COMPACT_LIMIT and last['value'] == out['value']
    with open(tmp_path / 'bench_detail.json') as fh:
        assert json.load(fh)['api']['note'] == out['api']['note']


def test_bounded_subprocess_ends_the_whole_process_group(tmp_path):
    """bench.run_bounded: a leg that does not finish within its limit is ended TOGETHER with the processes it spawned (the
    party processes of `-M3`), reports None, and honours the run's budget; a finished leg returns its output."""
    import time
    marker = tmp_path / 'child_alive'
    prog = ('import subprocess, sys, time\n'
            f'subprocess.Popen([sys.executable, "-c", "import time; time.sleep(4); open({str(marker)!r}, \'w\').write(\'x\')"])\n'
            'print("STARTED", flush=True)\ntime.sleep(60)\n')
    bench.LEG_BUDGET.update(deadline=None, leg_timeout=60.0)
    t0 = time.perf_counter()
    assert bench.run_bounded([sys.executable, '-c', prog], dict(os.environ), 1.5, cwd=str(tmp_path)) is None
    assert time.perf_counter() - t0 < 10
    time.sleep(4.5)
    assert not marker.exists()                       # the grandchild was ended with the group
    rc, out, err = bench.run_bounded([sys.executable, '-c', 'print("ok")'], dict(os.environ), 30, cwd=str(tmp_path))
    assert rc == 0 and out.strip() == 'ok'
    # budget: nothing starts when less than 3 s are left; a leg gets at most what is left
    bench.LEG_BUDGET.update(deadline=time.perf_counter() + 1.0)
    assert bench.run_bounded([sys.executable, '-c', 'print("ok")'], dict(os.environ), 30, cwd=str(tmp_path)) is None
    bench.LEG_BUDGET.update(deadline=None)
    a, b = bench.free_base_port(3), bench.free_base_port(3)
    assert a != b and 1024 < a < 65000
