"""HostView's device-resident integer algebra (mpyc_amd/finfields.py): what the reference's protocols do with
`.value` -- np_trunc, np_to_bits, np_from_bits, np_random_bits (runtime.py:838-873, 4391-4484, 4187-4273) --
restated on lazy views and compared with the SAME NumPy expressions on object arrays of Python integers:
  * handing a derived view to `field.array(...)` yields the reference's field elements (residues computed on the device),
  * every non-homomorphic step (`&`, `% 2^l`, `>>`, comparisons, anything unknown) yields the reference's integers,
    whether it runs on the device (exact views) or falls back to the true integers (derived views).
CPU: tests/cpuctx.py's Python-integer context; GPU (-m gpu): the kernels."""
import numpy as np
import pytest

PRIMES = [2**61 - 1, 2**64 - 189, 2**80 - 65, 2**96 - 17, 2**128 - 173, 2**31 - 1]


def _run(F, seed):
    from mpyc_amd.finfields import HostView
    p = F.modulus
    rng = np.random.default_rng(seed)
    n, f = 50, 12

    def rnd(shape):
        k = int(np.prod(shape))
        return np.array([int.from_bytes(rng.bytes(24), 'little') % p for _ in range(k)], dtype=object).reshape(shape)

    def lazy(a):
        return HostView(F.array(a), lazy=True)

    def same_field_array(view, want_ints):
        got = F.array(view)
        assert (np.asarray(got.value) == (want_ints % p)).all()

    # --- np_trunc (runtime.py:859-872) ---
    rb, av, rd = rnd((n * f,)), rnd((n,)), rnd((n,))
    l = p.bit_length() - 8
    want = np.sum(rb.reshape((n, f)) << np.arange(f), axis=1)
    got = np.sum(lazy(rb).reshape((n, f)) << np.arange(f), axis=1)
    assert isinstance(got, HostView) and got._is_lazy and not got._exact
    want = want.reshape((n,))
    got = got.reshape((n,))
    want += av
    got += lazy(av)
    e_w = want + (1 << l - 1) + (rd << f)
    e_g = got + (1 << l - 1) + (lazy(rd) << f)
    assert isinstance(e_g, HostView) and e_g._is_lazy
    same_field_array(e_g, e_w)
    c = rnd((n,))                                    # an opened value: canonical
    c_w = c & ((1 << f) - 1)
    c_g = lazy(c) & ((1 << f) - 1)
    assert isinstance(c_g, HostView) and c_g._exact and (np.asarray(c_g._real()) == c_w).all()
    same_field_array(got - c_g, want - c_w)
    assert (np.asarray((got - c_g)._real()) == (want - c_w)).all()          # the TRUE integers through the fallback

    # --- np_to_bits / np_from_bits (runtime.py:4417-4446, 4481-4484) ---
    lbits = 9
    shifts = np.arange(lbits)
    rbits = rnd((n, lbits))
    s_w = np.sum(rbits << shifts, axis=-1)
    s_g = np.sum(lazy(rbits) << shifts, axis=rbits.ndim - 1)
    same_field_array(s_g, s_w)
    same_field_array(((1 << 20) + (lazy(rd) << lbits) - s_g), ((1 << 20) + (rd << lbits) - s_w))
    cm_w = c % (1 << lbits)
    cm_g = lazy(c) % (1 << lbits)
    bits_w = np.int8(np.right_shift.outer(cm_w, shifts) & 1)
    bits_g = np.int8(np.right_shift.outer(cm_g, shifts) & 1)
    assert bits_g.dtype == np.int8 and (bits_g == bits_w).all()
    wide = np.arange(0, p.bit_length() + 3, 7)
    assert (np.int8(np.right_shift.outer(lazy(c), wide) & 1) == np.int8(np.right_shift.outer(c, wide) & 1)).all()
    assert (np.right_shift.outer(lazy(c), shifts)._real() == np.right_shift.outer(c, shifts)).all()     # fallback of the symbolic outer

    # --- np_random_bits (runtime.py:4249-4271) ---
    r, z = rnd((n,)), rnd((n,))
    r2_w = r**2 + z
    r2_g = lazy(r)**2 + lazy(z)
    same_field_array(r2_g, r2_w)
    opened = (r2_w % p)
    opened[3] = 0
    mask_g = lazy(opened) != 0
    assert isinstance(mask_g, np.ndarray) and mask_g.dtype == bool and (mask_g == (opened != 0)).all()
    assert np.count_nonzero(lazy(opened)) == np.count_nonzero(opened)
    assert (np.asarray(lazy(r)[mask_g]._real()) == r[opened != 0]).all()
    sq = rnd((n,))
    b_w = r * sq
    b_g = lazy(r) * lazy(sq)
    b_w %= p
    b_g %= p
    assert b_g._exact and (np.asarray(b_g._real()) == b_w).all()
    b_w += 1
    b_g += 1
    b_w *= (p + 1) >> 1
    b_g *= (p + 1) >> 1
    b_w <<= 3
    b_g <<= 3
    same_field_array(b_g, b_w)
    # raw class-level primitives on views stay on the device (runtime.py:4265)
    sq_in = lazy((r * r) % p)
    root = F.array._sqrt(sq_in, INV=False) if p % 4 == 3 else None
    if root is not None:
        assert isinstance(root, HostView) and root._is_lazy
        assert ((np.asarray(root._real()) ** 2 - (r * r)) % p == 0).all()

    # --- np_sgn (runtime.py:3646-3690): slices, transposes, vstack / cumsum of views and public arrays ---
    l = 7
    rb2 = rnd(((l + 1) * n,))
    shifts2 = np.arange(l - 1, -1, -1)
    cc = rnd((n,))
    rdl = rnd((n,))

    def sgn_expr(R, C, RD, wrap):
        s_sign = (R[-n:] << 1) - 1
        R = R[:l * n].reshape((n, l))
        r_modl = np.sum(R << shifts2, axis=1)
        a_r = wrap(av).reshape((n,)) + (1 << l) + r_modl
        opened_arg = a_r + (RD << l)
        c_ = C & ((1 << l) - 1)
        z_ = c_ - a_r
        c_bits = np.right_shift.outer(c_, shifts2).T & 1
        Rt = R.T
        Xor = c_bits + Rt - (c_bits * Rt << 1)
        zeros = np.zeros((1, n), dtype=object)
        ones = np.ones((1, n), dtype=object)
        SumXors = np.cumsum(np.vstack((zeros, Xor)), axis=0)
        e_ = s_sign - np.vstack((c_bits - Rt, ones)) + 3 * SumXors
        g_ = (np.arange(n) % 3 == 0)
        h_ = (1 - (g_ << 1)) * s_sign + 3
        z2 = z_ + (h_ << l - 1)
        return opened_arg, e_, 1 - Xor, z2, c_bits
    want5 = sgn_expr(rb2, cc, rdl, lambda x: x)
    got5 = sgn_expr(lazy(rb2), lazy(cc), lazy(rdl), lazy)
    for w_, g_ in zip(want5[:4], got5[:4]):
        assert isinstance(g_, HostView) and g_._is_lazy, type(g_)
        same_field_array(g_, w_)
        assert (g_._real() == w_).all()
    assert (np.asarray(got5[4]) == want5[4]).all()

    # --- _np_is_zero (runtime.py:3597-3616) and np_lsb (runtime.py:1817-1819): broadcasts, np.where ---
    kk = 5
    a1, r1, z1, u1 = rnd((n,)), rnd((kk, n)), (rnd((kk, n)) & 1), rnd((kk, n))
    c_w = a1 * r1 + (1 - (z1 << 1)) * u1
    c_g = lazy(a1).reshape((n,)) * lazy(r1).reshape((kk, n)) + (1 - (lazy(z1) << 1)) * lazy(u1)
    same_field_array(c_g, c_w)
    copen = c_w % p
    copen[0, :3] = 0
    zw = np.where(copen == 0, 0, z1)
    zg = np.where(lazy(copen) == 0, 0, lazy(z1))
    assert isinstance(zg, HostView) and zg._is_lazy
    same_field_array(zg, zw)
    sq_mask = (np.arange(kk * n).reshape((kk, n)) % 2 == 0)
    cw2 = np.where(sq_mask, 1 - zw, zw)
    cg2 = np.where(sq_mask, 1 - zg, zg)
    same_field_array(cg2, cw2)
    bw, rr = F.array(rnd((n,)) & 1), rnd((n,))
    lsb_w = F.array(av) + ((1 << 10) + (rr << 1) + np.asarray(bw.value))
    lsb_g = F.array(av) + ((1 << 10) + (lazy(rr) << 1) + HostView(bw, lazy=True))
    assert (np.asarray(lsb_g.value) == np.asarray(lsb_w.value)).all()
    x_w = np.where(c & 1, 1 - bw, bw)
    x_g = np.where(lazy(c) & 1, 1 - bw, bw)
    assert (np.asarray(x_g.value) == np.asarray(x_w.value)).all()

    # --- everything else falls back to the reference's integers ---
    v = lazy(c) + 5
    assert (np.asarray(v // 3) == (c + 5) // 3).all()
    assert (np.asarray(v >> 2) == (c + 5) >> 2).all()
    assert int(np.max(np.asarray(v._real()))) == int(np.max(c + 5))
    assert (np.asarray((lazy(c) << 70)._real()) == (c << 70)).all()          # beyond the limb width: true integers differ from residues
    assert (np.asarray((v & 0xff)) == ((c + 5) & 0xff)).all()               # `&` on a derived (non-exact) view
    import pickle
    assert (pickle.loads(pickle.dumps(v)) == (c + 5)).all()


@pytest.mark.parametrize('p', PRIMES)
def test_hostview_integer_algebra_host_logic(p, monkeypatch):
    from cpuctx import use_cpu_contexts
    import mpyc_amd.finfields as gff
    use_cpu_contexts(monkeypatch)
    monkeypatch.setattr(gff, '_ctx_cache', {})
    gff._pGF.cache_clear()
    try:
        _run(gff.GF(p), 7)
    finally:
        gff._pGF.cache_clear()


@pytest.mark.gpu
@pytest.mark.parametrize('p', PRIMES)
def test_hostview_integer_algebra_on_gpu(p):
    import mpyc_amd.finfields as gff
    _run(gff.GF(p), 11)
