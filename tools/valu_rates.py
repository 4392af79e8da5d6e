#!/usr/bin/env python3
"""Issue rates of the integer VALU by instruction kind and occupancy (ffgpu_valu_probe): wave64 instructions in 8 independent
dependent-chains per wave, 1 / 2 / 4 / 8 waves per SIMD -> lane-operations/s, shader clock under that load, lanes per cycle and
SIMD, cycles per wave instruction.  -> profiles/r0x_valu_rates.md"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mpyc_amd.engine import FieldContext

OPS = ['v_bitop3_b32 (VOP3, 3 sources)', 'v_add_u32 (VOP2)', 'v_mad_u64_u32 (VOP3b)', 'v_xor_b32 (VOP2)', 'v_perm_b32 (VOP3)',
       'v_lshrrev_b32 (VOP2, constant shift)', 'v_and_or_b32 (VOP3)', 'v_add3_u32 (VOP3)', 'v_mul_lo_u32 (VOP3)',
       'v_alignbit_b32 (VOP3, rotate)', 'v_lshl_or_b32 (VOP3)', 'v_alignbyte_b32 (VOP3)', 'v_lshrrev_b64 (VOP3, 64-bit shift)',
       'v_lshl_add_u64 (VOP3, 64-bit add)']
ctx = FieldContext(2**61 - 1)
simds = torch.cuda.get_device_properties(0).multi_processor_count * 4
print('| instruction | waves per SIMD | lane-ops/s | shader clock (MHz) | lanes per cycle and SIMD | cycles per wave64 instruction |')
print('|---|---|---|---|---|---|')
only = [int(v) for v in os.environ.get('VALU_OPS', '').split(',') if v]
for op, name in enumerate(OPS):
    if only and op not in only:
        continue
    for w in (1, 2, 4, 8):
        rate, mhz, _ = ctx.valu_probe(op, iters=4000, waves_per_simd=w)
        lpc = rate / (mhz * 1e6) / simds
        print(f'| `{name}` | {w} | {rate:.3e} | {mhz:.0f} | {lpc:.1f} | {64 / lpc:.2f} |')
