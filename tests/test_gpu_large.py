"""Maximum-size cases (sizes far beyond what the oracle can finish): arrays larger than
4 GiB, so every byte offset and, for GF(2^8), every element index crosses 2^32.  Whole-array checks use
size-independent properties evaluated on the device (round trip, commutativity, linearity of the sum);
windows at the start, across the 2^32 boundary and at the ragged tail are compared with the oracle."""
import numpy as np
import pytest

from oracle import pyoracle as po
from fieldutil import unpack

pytestmark = pytest.mark.gpu
torch = pytest.importorskip('torch')

P61 = 2**61 - 1


@pytest.fixture(scope='module')
def eng():
    assert torch.cuda.is_available()
    from mpyc_amd import engine
    return engine


def windows(n, cross):
    w = [(0, 64), (n - 64, n)]
    if 64 < cross < n - 64:
        w.append((cross - 32, cross + 32))
    return w


def window_ints(arr, lo, hi, eb):
    t = arr.t[lo:hi].cpu().numpy()
    t = t.view(np.uint64) if eb >= 8 else t.view(np.uint32) if eb == 4 else t
    return unpack(t, eb)


def free_gib():
    free, _ = torch.cuda.mem_get_info(0)
    return free / 2**30


def test_gf256_beyond_4gib_elements(eng):
    if free_gib() < 40:
        pytest.skip('needs 40 GiB of free HBM')
    n = 2**32 + 4099                                            # element INDEX crosses 2^32; ragged tail
    F = po.Field(0x11b, True)
    ctx = eng.FieldContext(0x11b, True, device=0)
    g = torch.Generator(device='cuda:0').manual_seed(1)
    a = eng.DevArray(ctx, torch.randint(0, 256, (n,), dtype=torch.uint8, device='cuda:0', generator=g), n)
    b = eng.DevArray(ctx, torch.randint(0, 256, (n,), dtype=torch.uint8, device='cuda:0', generator=g), n)
    c = ctx.mul(a, b)
    for lo, hi in windows(n, 2**32):
        x, y, z = (window_ints(v, lo, hi, 1) for v in (a, b, c))
        assert z == [po.mul(F, p, q) for p, q in zip(x, y)], (lo, hi)
    assert torch.equal(ctx.mul(b, a).t, c.t)                     # commutativity over the whole array
    sh = ctx.split_rng(c, 1, 3, key=bytes(range(32)))
    lam = po.recombination_vector(F, [1, 3], 0)
    back = ctx.recombine([sh.row(0), sh.row(2)], lam)
    assert torch.equal(back.t, c.t)                              # encode -> erase a row -> decode
    assert not torch.equal(sh.row(0).t, c.t)
    del sh, back
    s = ctx.sbox(a, *sbox_rows())
    tab = sbox_table()
    for lo, hi in windows(n, 2**32):
        assert window_ints(s, lo, hi, 1) == [tab[v] for v in window_ints(a, lo, hi, 1)]


def sbox_rows():
    import json
    import os
    g = json.load(open(os.path.join(os.path.dirname(__file__), 'golden', 'sbox.json')))
    return g['rows8'], g['b']


def sbox_table():
    import json
    import os
    return json.load(open(os.path.join(os.path.dirname(__file__), 'golden', 'sbox.json')))['table']


def test_p61_beyond_4gib_bytes(eng):
    if free_gib() < 60:
        pytest.skip('needs 60 GiB of free HBM')
    n = 2**29 + 2**27 + 5                                        # 5 GiB per array: byte offsets cross 2^32
    F = po.Field(P61, False)
    ctx = eng.FieldContext(P61, False, device=0)
    g = torch.Generator(device='cuda:0').manual_seed(2)
    raw = lambda: torch.randint(0, P61, (n,), dtype=torch.int64, device='cuda:0', generator=g)
    a, b = eng.DevArray(ctx, raw(), n), eng.DevArray(ctx, raw(), n)
    c = ctx.mul(a, b)
    for lo, hi in windows(n, 2**29):
        x, y, z = (window_ints(v, lo, hi, 8) for v in (a, b, c))
        assert z == [p * q % P61 for p, q in zip(x, y)], (lo, hi)
    # fused gate: product inside share generation, device CSPRNG, m = 3, t = 1; any 2 rows decode
    sh = ctx.split_rng(a, 1, 3, key=bytes(range(32)), mul_by=b)
    for xs in ([1, 2], [2, 3]):
        lam = po.recombination_vector(F, xs, 0)
        back = ctx.recombine([sh.row(x - 1) for x in xs], lam)
        assert torch.equal(back.t, c.t), xs
        del back
    # the three rows lie on a line: row1 - 2 row2 + row3 == 0 (degree check, whole array)
    z = ctx.recombine([sh.row(0), sh.row(1), sh.row(2)], [1, P61 - 2, 1])
    assert not z.t.any()
    del sh, z
    # checksum of checksums: sum is linear
    sa, sb, sab = (ctx.sum(v).to_ints()[0] for v in (a, b, ctx.add(a, b)))
    assert (sa + sb) % P61 == sab
    assert ctx.dot(a, b).to_ints()[0] == ctx.sum(c).to_ints()[0]


def test_valu_probe_reports_plausible_rates():
    """ffgpu_valu_probe (the compute-side yardstick of bench.py): every instruction kind, one and four waves per SIMD -- rates
    and clocks in the range an MI355X can produce, two-operand instructions faster than three-operand ones at full occupancy,
    a lone wave slower than a full SIMD."""
    from mpyc_amd.engine import FieldContext
    ctx = FieldContext(2**61 - 1)
    rates = {}
    for op in range(14):
        for w in (1, 4):
            rate, mhz, cyc = ctx.valu_probe(op, iters=1000, waves_per_simd=w)
            assert 5e12 < rate < 1.5e14 and 500 < mhz < 3500 and cyc > 0, (op, w, rate, mhz, cyc)
            rates[op, w] = rate
    assert rates[3, 4] > 1.3 * rates[0, 4]          # v_xor_b32 (VOP2) against v_bitop3_b32 (VOP3)
    assert rates[1, 4] > 1.5 * rates[1, 1]          # a SIMD needs two or more waves for the two-operand rate
    assert rates[1, 4] > 1.3 * rates[9, 4]          # v_alignbit_b32 (the ChaCha rotate) is an ordinary three-operand instruction
    for op64 in (12, 13):                           # 64-bit shift / add (the carry passes of the digit arithmetic): the same class,
        assert 0.7 * rates[2, 4] < rates[op64, 4] < 1.4 * rates[2, 4]        # not a multi-pass instruction (v_mad_u64_u32 beside them)
    with pytest.raises(ValueError):
        ctx.valu_probe(14)
