// hostcheck.cpp -- TEST-ONLY harness.  Compiles mpyc_amd/csrc/fields.hpp with
// plain g++ (no HIP) so that the exact arithmetic the GPU kernels run can be
// compared against Python integers on a machine without a GPU.  It is built
// into tests/_hostcheck.so by tests/conftest.py and is never loaded by the
// product package; libffgpu.so has no CPU execution path.
#include <stdint.h>
#include <string.h>
#include <type_traits>
#include "../mpyc_amd/csrc/policy_build.hpp"
#include "../mpyc_amd/csrc/rng.hpp"
#include "../mpyc_amd/csrc/bitslice.hpp"

using namespace ffgpu;

enum { HC_ADD = 0, HC_SUB = 1, HC_MUL = 2, HC_NEG = 3, HC_REDUCE = 4, HC_MULADD = 5, HC_MULADD_SMALL = 6, HC_DOT = 7, HC_SHARE = 8, HC_LAZY = 9, HC_COLDOT = 10, HC_LDOT = 11, HC_CHAIN = 12 };

template <class F>
static typename F::word ldw(const unsigned char* p, size_t i) {
    typename F::elem e;
    memcpy(&e, p + i * sizeof(e), sizeof(e));
    if constexpr (sizeof(typename F::elem) == 12) {
        typename F::word w;
        w.lo = (uint64_t)e.x[0] | ((uint64_t)e.x[1] << 32);
        w.hi = e.x[2];
        return w;
    } else if constexpr (F::EPW == 1) {
        return e;
    } else {
        return (typename F::word)e;
    }
}
template <class F>
static void stw(unsigned char* p, size_t i, typename F::word w) {
    typename F::elem e;
    if constexpr (sizeof(typename F::elem) == 12) {
        e.x[0] = (uint32_t)w.lo;
        e.x[1] = (uint32_t)(w.lo >> 32);
        e.x[2] = (uint32_t)w.hi;
    } else if constexpr (F::EPW == 1) {
        e = w;
    } else {
        e = (typename F::elem)w;
    }
    memcpy(p + i * sizeof(e), &e, sizeof(e));
}

template <class F>
static typename F::word cst(const F& f, const uint64_t* l) {
    if constexpr (sizeof(typename F::word) == 24) {
        typename F::word w;
        w.lo = l[0];
        w.mid = l[1];
        w.hi = l[2];
        return w;
    } else if constexpr (sizeof(typename F::word) == 16) {
        typename F::word w;
        w.lo = l[0];
        w.hi = l[1];
        return w;
    } else if constexpr (F::EPW == 4) {
        return (uint32_t)(l[0] & 0xff) * 0x01010101u;
    } else {
        return (typename F::word)l[0];
    }
}

template <class F, class = void>
struct has_lacc : std::false_type {};
template <class F>
struct has_lacc<F, std::void_t<typename F::lacc> > : std::true_type {};

template <class F, class = void>
struct has_chain : std::false_type {};
template <class F>
struct has_chain<F, std::void_t<decltype(F::CHAIN_MAX_NL)> > : std::true_type {};

// a,b,c: n elements each; for HC_DOT: a holds k rows of n elements, lam holds k
// canonical 2-limb constants, x ignored.  For HC_MULADD_SMALL x is the small
// public multiplier.
template <class F>
static int run(const PolicyBlob& pb, int op, const unsigned char* a, const unsigned char* b,
               const unsigned char* c, unsigned char* out, size_t n, uint32_t x, const uint64_t* lam, int k) {
    F f;
    memcpy(&f, pb.bytes, sizeof(F));
    for (size_t i = 0; i < n; ++i) {
        typename F::word r;
        switch (op) {
            case HC_ADD: r = f.add(ldw<F>(a, i), ldw<F>(b, i)); break;
            case HC_SUB: r = f.sub(ldw<F>(a, i), ldw<F>(b, i)); break;
            case HC_MUL: r = f.mul(ldw<F>(a, i), ldw<F>(b, i)); break;
            case HC_NEG: r = f.neg(ldw<F>(a, i)); break;
            case HC_REDUCE: r = f.reduce_raw(ldw<F>(a, i)); break;
            case HC_MULADD: r = f.muladd(ldw<F>(a, i), ldw<F>(b, i), ldw<F>(c, i)); break;
            case HC_MULADD_SMALL: r = f.muladd_small(ldw<F>(a, i), x, ldw<F>(c, i)); break;
            case HC_DOT: {
                typename F::acc s;
                f.acc_zero(s);
                for (int j = 0; j < k; ++j)
                    f.acc_mac(s, f.prep(cst<F>(f, lam + (sizeof(typename F::word) == 24 ? 3 : 2) * j)), ldw<F>(a, (size_t)j * n + i));
                r = f.acc_reduce(s);
                break;
            }
            case HC_CHAIN: {
                // a product chain in digits (fields.hpp DigitChain; kernels.hpp ff_pow_digits): u v, then six times
                // square-and-multiply, partially reduced throughout, canonical at the end.  x = number of digits (3..7)
                if constexpr (has_chain<F>::value) {
                    auto go = [&](auto nl) -> int {
                        constexpr int NL = decltype(nl)::value;
                        DigitChain<NL> dc;
                        if (!f.template chain_setup<NL>(dc)) return 3;
                        auto u = f.template chain_in<NL>(dc, ldw<F>(a, i)), v = f.template chain_in<NL>(dc, ldw<F>(b, i));
                        auto t = dc.mul(u, v);
                        for (int j = 0; j < 6; ++j) {
                            t = dc.sqr(t);
                            t = dc.mul(t, (j & 1) ? u : v);
                        }
                        r = f.template chain_out<NL>(dc, t);
                        return 0;
                    };
                    int rc = 3;
                    auto pick = [&](auto nl) {
                        constexpr int NL = decltype(nl)::value;
                        if constexpr (NL >= F::CHAIN_MIN_NL && NL <= F::CHAIN_MAX_NL) {
                            if ((int)x == NL) rc = go(nl);
                        }
                    };
                    pick(std::integral_constant<int, 3>()); pick(std::integral_constant<int, 4>()); pick(std::integral_constant<int, 5>());
                    pick(std::integral_constant<int, 6>()); pick(std::integral_constant<int, 7>());
                    if (rc) return rc;
                } else {
                    return 2;
                }
                break;
            }
            case HC_LDOT: {
                // the recombination kernels' dot product in 28-bit digits (fields.hpp LazyDot), policies that have one
                if constexpr (has_lacc<F>::value) {
                    typename F::lacc s;
                    f.lacc_zero(s);
                    for (int j = 0; j < k; ++j)
                        f.lacc_mac(s, f.prep(cst<F>(f, lam + (sizeof(typename F::word) == 24 ? 3 : 2) * j)), ldw<F>(a, (size_t)j * n + i));
                    r = f.lacc_reduce(s);
                } else {
                    return 2;
                }
                break;
            }
            case HC_COLDOT: {
                // the skinny products' column accumulators (fields.hpp ColAcc): lam = the shared operand, split into limbs once
                if constexpr (col_mac_ok<F>::value) {
                    ColAcc<typename F::acc> ca;
                    ca.zero();
                    int cnt = 0;
                    for (int j = 0; j < k; ++j) {
                        ca.mac(col_limbs(f.prep(cst<F>(f, lam + 2 * j))), ldw<F>(a, (size_t)j * n + i));
                        if (++cnt >= 192) {                     // the kernels' flush: reduce, the residue re-enters as ONE term (1 x residue)
                            const typename F::word part = f.acc_reduce(ca.gather());
                            ca.zero();
                            ca.c00 = (uint32_t)part;
                            ca.c01 = (uint32_t)(part >> 32);
                            cnt = 1;
                        }
                    }
                    r = f.acc_reduce(ca.gather());
                } else {
                    return 2;
                }
                break;
            }
            case HC_SHARE: {
                // a: secrets (n), c: k rows of n coefficients, x: number of parties m; out: the share of party x, reached
                // by x forward-difference steps from f(0) = s (fields.hpp share_diff_*; the kernels' share loop)
                typename F::word cc[4], dd[4];
                for (int j = 0; j < k && j < 4; ++j) cc[j] = ldw<F>(c, (size_t)j * n + i);
                r = ldw<F>(a, i);
                if constexpr (!F::BINARY) {
                    auto walk = [&](auto tc) {
                        constexpr int T = decltype(tc)::value;
                        typename F::word c_[T], d_[T];
                        for (int j = 0; j < T; ++j) c_[j] = cc[j];
                        share_diff_init<F, T>(f, c_, d_);
                        for (uint32_t pt = 1; pt <= x; ++pt) r = share_diff_next<F, T>(f, r, d_);
                    };
                    switch (k) {
                        case 1: walk(std::integral_constant<int, 1>()); break;
                        case 2: walk(std::integral_constant<int, 2>()); break;
                        case 3: walk(std::integral_constant<int, 3>()); break;
                        case 4: walk(std::integral_constant<int, 4>()); break;
                        default: return 1;
                    }
                }
                (void)dd;
                break;
            }
            case HC_LAZY: {
                // the product chains of ff_pow / k_inv_batch (kernels.hpp): partially reduced intermediates, one canon at
                // the end.  x != 0: the operands 0, 1, 2 enter as p, p + 1, p + 2 (the largest values a chain can carry).
                if constexpr (std::is_same<F, PM64<false, true> >::value || std::is_same<F, PM64<false, false> >::value ||
                              std::is_same<F, PM64<true, false> >::value) {
                    typename F::word u = ldw<F>(a, i), v = ldw<F>(b, i);
                    if (x && std::is_same<F, PM64<false, true> >::value) {
                        if (u <= 2) u += f.p;
                        if (v <= 2) v += f.p;
                    }
                    if (x && std::is_same<F, PM64<true, false> >::value) {      // p = 2^64 - c: every u < c has a second 64-bit representative
                        if (u < f.c) u += f.p;
                        if (v < f.c) v += f.p;
                    }
                    r = f.mul_lazy(u, v);
                    for (int j = 0; j < 6; ++j) {
                        r = f.sqr_lazy(r);
                        r = f.mul_lazy(r, (j & 1) ? u : v);
                    }
                    r = f.canon(r);
                } else {
                    return 2;
                }
                break;
            }
            default: return 1;
        }
        stw<F>(out, i, r);
    }
    return 0;
}

extern "C" int hc_run(int binary, const uint64_t* modulus, int nlimbs, int op, const unsigned char* a,
                      const unsigned char* b, const unsigned char* c, unsigned char* out, size_t n,
                      uint32_t x, const uint64_t* lam, int k, int* policy_kind, int* elem_bytes) {
    PolicyBlob pb;
    memset(&pb, 0, sizeof(pb));
    int rc = binary ? build_binary_policy(&pb, modulus, nlimbs) : build_prime_policy3(&pb, modulus, nlimbs);
    if (rc) return 100 + rc;
    if (policy_kind) *policy_kind = pb.kind;
    if (elem_bytes) *elem_bytes = pb.elem_bytes;
    if (n == 0) return 0;
    switch (pb.kind) {
        case POL_PM64_MERSENNE: return run<PM64<false, true> >(pb, op, a, b, c, out, n, x, lam, k);
        case POL_PM64_K64: return run<PM64<true, false> >(pb, op, a, b, c, out, n, x, lam, k);
        case POL_PM64_GEN: return run<PM64<false, false> >(pb, op, a, b, c, out, n, x, lam, k);
        case POL_RC64: return run<RC64>(pb, op, a, b, c, out, n, x, lam, k);
        case POL_RC32: return run<RC32>(pb, op, a, b, c, out, n, x, lam, k);
        case POL_PM128_K128: return run<PM128<true> >(pb, op, a, b, c, out, n, x, lam, k);
        case POL_PM128_GEN: return run<PM128<false> >(pb, op, a, b, c, out, n, x, lam, k);
        case POL_PM96: return run<PM96>(pb, op, a, b, c, out, n, x, lam, k);
        case POL_MONT128: return run<MONT128>(pb, op, a, b, c, out, n, x, lam, k);
        case POL_GF2P8: return run<GF2P8>(pb, op, a, b, c, out, n, x, lam, k);
        case POL_GF2W32: return run<GF2W32>(pb, op, a, b, c, out, n, x, lam, k);
        case POL_GF2W64: return run<GF2W64>(pb, op, a, b, c, out, n, x, lam, k);
        case POL_GF2W128: return run<GF2W128>(pb, op, a, b, c, out, n, x, lam, k);
        case POL_PM192: return run<PM192>(pb, op, a, b, c, out, n, x, lam, k);
        case POL_MONT192: return run<MONT192>(pb, op, a, b, c, out, n, x, lam, k);
        default: return 2;
    }
}

// ---- device CSPRNG (rng.hpp) on the host ---------------------------------------------------
extern "C" void hc_chacha_block(const uint32_t key[8], const uint32_t w[4], int rounds, uint32_t out[16]) {
    chacha_block(key, w[0], w[1], w[2], w[3], rounds, out);
}

template <class F, int T>
static void draw_rows(const PolicyBlob& pb, const RngKey& rk, unsigned char* out, size_t cstride, size_t n, int row0) {
    F f;
    memcpy(&f, pb.bytes, sizeof(F));
    uint64_t R[2];
    rng_const(pb, R);
    constexpr int WPP = sizeof(typename F::word) >= 16 ? 1 : 16 / sizeof(typename F::word);     // words per pack (kernels.hpp Pack<W>::N)
    constexpr int EPV = WPP * F::EPW;
    size_t npacks = (n + EPV - 1) / EPV;
    for (size_t i = 0; i < npacks; ++i) {
        typename F::word c[T][WPP];
        rng_draw_pack<F, T, WPP>(f, rk, R[0], R[1], (uint64_t)i, (uint64_t)npacks, c);
        for (int j = 0; j < T; ++j)
            for (int q = 0; q < WPP; ++q)
                for (int b = 0; b < F::EPW; ++b) {
                    size_t e = i * EPV + (size_t)q * F::EPW + b;
                    if (e >= n) continue;
                    typename F::word v = c[j][q];
                    if constexpr (F::EPW > 1) v = (typename F::word)((v >> (8 * b)) & 0xffu);
                    stw<F>(out + (size_t)(row0 + j) * cstride * sizeof(typename F::elem), e, v);
                }
    }
}

template <class F>
static int rng_rows(const PolicyBlob& pb, RngKey rk, int t, unsigned char* out, size_t cstride, size_t n) {
    switch (t) {
        case 1: draw_rows<F, 1>(pb, rk, out, cstride, n, 0); return 0;
        case 2: draw_rows<F, 2>(pb, rk, out, cstride, n, 0); return 0;
        case 3: draw_rows<F, 3>(pb, rk, out, cstride, n, 0); return 0;
        case 4: draw_rows<F, 4>(pb, rk, out, cstride, n, 0); return 0;
        default:
            for (int j = 0; j < t; ++j) {
                RngKey rj = rk;
                rj.nonce[1] += (uint32_t)(j + 1);
                draw_rows<F, 1>(pb, rj, out, cstride, n, j);
            }
            return 0;
    }
}

extern "C" int hc_rng_coeffs(int binary, const uint64_t* modulus, int nlimbs, const unsigned char* key32,
                             uint64_t nonce, int rounds, int t, unsigned char* out, size_t cstride, size_t n) {
    PolicyBlob pb;
    memset(&pb, 0, sizeof(pb));
    int rc = binary ? build_binary_policy(&pb, modulus, nlimbs) : build_prime_policy3(&pb, modulus, nlimbs);
    if (rc) return 100 + rc;
    RngKey rk;
    memset(&rk, 0, sizeof(rk));
    memcpy(rk.key, key32, 32);
    rk.nonce[0] = (uint32_t)nonce;
    rk.nonce[1] = (uint32_t)(nonce >> 32);
    rk.rounds = rounds ? (uint32_t)rounds : 20u;
    switch (pb.kind) {
        case POL_PM64_MERSENNE: return rng_rows<PM64<false, true> >(pb, rk, t, out, cstride, n);
        case POL_PM64_K64: return rng_rows<PM64<true, false> >(pb, rk, t, out, cstride, n);
        case POL_PM64_GEN: return rng_rows<PM64<false, false> >(pb, rk, t, out, cstride, n);
        case POL_RC64: return rng_rows<RC64>(pb, rk, t, out, cstride, n);
        case POL_RC32: return rng_rows<RC32>(pb, rk, t, out, cstride, n);
        case POL_PM128_K128: return rng_rows<PM128<true> >(pb, rk, t, out, cstride, n);
        case POL_PM128_GEN: return rng_rows<PM128<false> >(pb, rk, t, out, cstride, n);
        case POL_PM96: return rng_rows<PM96>(pb, rk, t, out, cstride, n);
        case POL_MONT128: return rng_rows<MONT128>(pb, rk, t, out, cstride, n);
        case POL_GF2P8: return rng_rows<GF2P8>(pb, rk, t, out, cstride, n);
        case POL_GF2W32: return rng_rows<GF2W32>(pb, rk, t, out, cstride, n);
        case POL_GF2W64: return rng_rows<GF2W64>(pb, rk, t, out, cstride, n);
        case POL_GF2W128: return rng_rows<GF2W128>(pb, rk, t, out, cstride, n);
        case POL_PM192: return rng_rows<PM192>(pb, rk, t, out, cstride, n);
        case POL_MONT192: return rng_rows<MONT192>(pb, rk, t, out, cstride, n);
        default: return 2;
    }
}

// digits of the matrix-core product's operand representative (fields.hpp limb_digits), L = 4 or 8
extern "C" int hc_limb_digits(uint64_t x, uint64_t p, int L, int8_t* out) {
    if (L == 8) {
        int8_t d[8];
        limb_digits<8>(x, p, d);
        memcpy(out, d, 8);
    } else if (L == 4) {
        int8_t d[4];
        limb_digits<4>(x, p, d);
        memcpy(out, d, 4);
    } else {
        return 1;
    }
    return 0;
}

extern "C" int hc_limb_digits_wide(const uint64_t* x2, const uint64_t* p2, int L, int8_t* out) {
    if (L == 16) {
        int8_t d[16];
        limb_digits_wide<16>(x2[0], x2[1], p2[0], p2[1], d);
        memcpy(out, d, 16);
    } else if (L == 12) {
        int8_t d[12];
        limb_digits_wide<12>(x2[0], x2[1], p2[0], p2[1], d);
        memcpy(out, d, 12);
    } else {
        return 1;
    }
    return 0;
}


// the bit-sliced GF(2^64) product of k_gf2w64_mul_bitsliced on the 16 elements of a lane (bitslice.hpp mul16_packed: one
// transpose per operand, packed Karatsuba on bit-planes, fold modulo x^64 + x^4 + x^3 + x + 1, transpose back)
extern "C" int hc_bs64_mul16_packed(const uint64_t* a, const uint64_t* b, uint64_t* out) {
    uint32_t alo[16], ahi[16], blo[16], bhi[16], olo[16], ohi[16];
    for (int e = 0; e < 16; ++e) {
        alo[e] = (uint32_t)a[e]; ahi[e] = (uint32_t)(a[e] >> 32);
        blo[e] = (uint32_t)b[e]; bhi[e] = (uint32_t)(b[e] >> 32);
    }
    bs64::mul16_packed(alo, ahi, blo, bhi, olo, ohi);
    for (int e = 0; e < 16; ++e) out[e] = (uint64_t)olo[e] | ((uint64_t)ohi[e] << 32);
    return 0;
}
// one 32 x 32 bit transpose (out word i, bit e = in word e, bit i)
extern "C" int hc_bs64_transpose32(const uint32_t* in, uint32_t* out) {
    uint32_t A[32];
    for (int i = 0; i < 32; ++i) A[i] = in[i];
    bs64::transpose32(A);
    for (int i = 0; i < 32; ++i) out[i] = A[i];
    return 0;
}
