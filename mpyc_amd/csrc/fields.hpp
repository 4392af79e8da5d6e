// fields.hpp -- field-arithmetic policies used by every kernel in libffgpu.
//
// Each policy is a small POD passed BY VALUE as a kernel argument, so its
// constants (modulus, fold constant, reciprocal, shift) are wave-uniform and
// live in SGPRs; there is no table to stage in LDS for these fields.
//
// A policy F provides
//     F::elem            addressable storage unit in HBM (uint8/32/64, u128e)
//     F::word            unit the arithmetic works on (== elem, except the
//                        packed GF(2^n<=8) policy where a word is 4 elements)
//     F::EPW             elements per word
//     add/sub/neg/mul    canonical in, canonical out
//     reduce_raw         arbitrary bit pattern -> canonical
//     muladd_small(y,x,c)  y*x + c for a 32-bit public x (Horner step of
//                        share generation, thresha.py:41-43 / :61-63)
//     acc / acc_zero / acc_mac / acc_reduce
//                        unreduced dot-product accumulation with ONE final
//                        reduction (finfields.py:1126-1135 object matmul)
//
// The functions are __host__ __device__ so that tests/hostcheck.cpp can compile
// this very file with g++ and compare it against Python integers without a GPU.
// That harness is test-only; the shipped library has no CPU code path.
#pragma once
#include <stdint.h>
#include <stddef.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define FF_HD __host__ __device__ __forceinline__
#else
#define FF_HD inline
#endif

typedef unsigned __int128 ff_u128;

struct alignas(16) u128e {
    uint64_t lo, hi;
};

FF_HD ff_u128 ff_make128(uint64_t hi, uint64_t lo) { return ((ff_u128)hi << 64) | lo; }
FF_HD uint64_t ff_lo(ff_u128 x) { return (uint64_t)x; }
FF_HD uint64_t ff_hi(ff_u128 x) { return (uint64_t)(x >> 64); }

// ---------------------------------------------------------------------------
// PM64: prime p = 2^k - c, 33 <= k <= 64, c < 2^min(31,(k-1)/2).
// All default MPyC primes are of this shape (find_prime_root = largest Blum
// prime below 2^l, finfields.py:311-344): 2^61-1, 2^64-189, ...
// Reduction = fold the high part down with 2^k == c (mod p); no division.
// ---------------------------------------------------------------------------
template <bool K64, bool C1>
struct PM64 {
    typedef uint64_t elem;
    typedef uint64_t word;
    enum { EPW = 1 };
    enum { BINARY = 0 };
    uint64_t p;     // modulus
    uint64_t mask;  // 2^k - 1
    uint32_t c;     // 2^k - p
    uint32_t k;

    struct acc {
        uint64_t a0, a1, a2;
    };
    // constants are used as-is (no domain conversion)
    FF_HD uint64_t prep(uint64_t cst) const { return cst; }


    FF_HD uint64_t csub(uint64_t x) const { return x >= p ? x - p : x; }

    FF_HD uint64_t add(uint64_t a, uint64_t b) const {
        if (K64) {
            // p = 2^64 - c: a + b - p = a + b + c - 2^64.  Both "the sum wrapped" and "the sum is >= p" show as a CARRY
            // (of a + b, resp. of (a + b) + c), which the adds produce for free -- no 64-bit compares, no subtraction:
            // 4 add instructions + 2 selects (the share loop of k_split is 3 such additions per party and secret)
            uint64_t s, t;
            const bool c1 = __builtin_add_overflow(a, b, &s);
            const bool c2 = __builtin_add_overflow(s, (uint64_t)c, &t);
            return (c1 | c2) ? t : s;
        }
        return csub(a + b);
    }
    FF_HD uint64_t sub(uint64_t a, uint64_t b) const {
        uint64_t d;
        const bool borrow = __builtin_sub_overflow(a, b, &d);
        return borrow ? d + p : d;
    }
    FF_HD uint64_t neg(uint64_t a) const { return a ? p - a : 0; }

    // x < 2^(2k) when k < 64; any 128-bit x when k == 64.
    FF_HD uint64_t red128(ff_u128 x) const {
        if (K64) {
            ff_u128 t = (ff_u128)ff_hi(x) * c + ff_lo(x);  // < 2^96
            uint64_t tl = ff_lo(t);
            uint64_t u = tl + ff_hi(t) * (uint64_t)c;  // hi(t) <= 2^31
            if (u < tl) u += c;                        // wrapped: 2^64 == c
            return csub(u);
        }
        return csub(fold128(x));
    }
    // 33 <= k <= 63 and x < 2^(2k+1): x mod p up to one multiple of p, value <= 2^k + 1 for c == 1 (< 2p otherwise).
    // The 128-bit shifts by k are written on the two words (64 - k is in 1..31): a generic `x >> k` makes the compiler
    // guard k >= 64 and k == 0 with selects, ~10 instructions of the ~45 a product costs -- the exponentiation-bound
    // kernels (inverse, square root, pow) are VALU-bound on exactly this.
    FF_HD uint64_t fold128(ff_u128 x) const {
        const uint32_t s = 64 - k;
        uint64_t xh = (ff_hi(x) << s) | (ff_lo(x) >> k);
        uint64_t xl = ff_lo(x) & mask;
        if (C1) {
            uint64_t w = xl + xh;  // < 2^(k+2)
            return (w & mask) + (w >> k);
        }
        ff_u128 w = (ff_u128)xh * c + xl;
        uint64_t wh = (ff_hi(w) << s) | (ff_lo(w) >> k);
        return (ff_lo(w) & mask) + wh * (uint64_t)c;  // < 2p
    }

    // p = 2^k - 1 (C1; 33 <= k <= 61, i.e. 2^61 - 1): the product is never formed as a 128-bit number.  With
    // a = a1 2^32 + a0 and 2^k == 1:   a b == p11 2^(64-k) + (m >> (k-32)) + ((m mod 2^(k-32)) << 32) + (p00 mod 2^k) + (p00 >> k),
    // p00 = a0 b0, m = a0 b1 + a1 b0, p11 = a1 b1 -- four v_mad_u64_u32 whose 64-bit results are used as they are (the
    // schoolbook 128-bit product chains them through carries and register shuffles), one shift per term, plain adds.
    // Operands <= 2^k + 2 (PARTIALLY reduced) give S < 2^(k+2); fold(S) <= 2^k + 2 again, so chains of products
    // (ff_pow, the batched inverse) subtract p once at the end (`canon`): ~21 VALU issue slots per product instead of
    // the 45 of the generic 2^k - c path below.
    FF_HD uint64_t presum(uint64_t a, uint64_t b) const {
        const uint32_t a0 = (uint32_t)a, a1 = (uint32_t)(a >> 32), b0 = (uint32_t)b, b1 = (uint32_t)(b >> 32);
        const uint64_t p00 = (uint64_t)a0 * b0;
        const uint64_t m = (uint64_t)a0 * b1 + (uint64_t)a1 * b0;      // < 2^(k+2)
        const uint64_t p11 = (uint64_t)a1 * b1;                        // <= 2^(2k-64)
        return presum_terms<0>(p00, m, p11);
    }
    FF_HD uint64_t presum_sqr(uint64_t a) const {
        const uint32_t a0 = (uint32_t)a, a1 = (uint32_t)(a >> 32);
        return presum_terms<1>((uint64_t)a0 * a0, (uint64_t)a0 * a1, (uint64_t)a1 * a1);     // middle term 2 a0 a1
    }
    // x mod 2^k for k >= 33: the low word passes, one AND on the high word (with `x & mask` the compiler cannot know
    // that the low word of the run-time mask is all ones)
    FF_HD uint64_t low_k(uint64_t x) const {
        return ((uint64_t)((uint32_t)(x >> 32) & (uint32_t)(mask >> 32)) << 32) | (uint32_t)x;
    }
    template <int DBL = 0>
    FF_HD uint64_t presum_terms(uint64_t p00, uint64_t m, uint64_t p11) const {          // the middle term is m << DBL
        const uint32_t kk = k - 32;
        uint64_t s = (p11 << (64 - k)) + (m >> (kk - DBL));
        s += (uint64_t)(((uint32_t)m << DBL) & (uint32_t)(mask >> 32)) << 32;
        s += low_k(p00);
        s += p00 >> k;
        return s;
    }
    FF_HD uint64_t fold64(uint64_t s) const { return low_k(s) + (s >> k); }

    FF_HD uint64_t mul(uint64_t a, uint64_t b) const {
        if (!K64 && C1) return csub(fold64(presum(a, b)));
        return red128((ff_u128)a * b);
    }
    // k == 64: ANY 64-bit word is a valid representative (x and x - p when x >= p), and the fold below keeps a product inside
    // 64 bits without the final conditional subtraction -- chains of products (ff_pow, the batched inverse) skip the 64-bit
    // compare + subtract + two selects per product (6 of ~40 instructions) and subtract p once at the end (`canon`)
    FF_HD uint64_t red128_lazy(ff_u128 x) const {
        ff_u128 t = (ff_u128)ff_hi(x) * c + ff_lo(x);  // < 2^96
        uint64_t tl = ff_lo(t);
        uint64_t u = tl + ff_hi(t) * (uint64_t)c;      // hi(t) <= 2^31, c < 2^31
        if (u < tl) u += c;                            // wrapped: 2^64 == c; u < 2^62 here, no second wrap
        return u;
    }
    FF_HD uint64_t mul_lazy(uint64_t a, uint64_t b) const {
        if (!K64 && C1) return fold64(presum(a, b));
        if (K64) return red128_lazy((ff_u128)a * b);
        return mul(a, b);
    }
    FF_HD uint64_t sqr_lazy(uint64_t a) const {
        if (!K64 && C1) return fold64(presum_sqr(a));
        if (K64) return red128_lazy((ff_u128)a * a);
        return mul(a, a);
    }
    FF_HD uint64_t canon(uint64_t x) const { return ((!K64 && C1) || K64) ? csub(x) : x; }
    FF_HD uint64_t reduce_raw(uint64_t x) const {
        if (K64) return csub(x);
        return red128((ff_u128)x);
    }
    // Horner step y*x + cadd with a 32-bit public x: the value T has only 96 bits
    // (hi < 2^32, lo), so the fold is written out on (hi, lo) instead of going
    // through the generic 128-bit path: 3 v_mad_u64_u32 + a few adds for k == 64.
    FF_HD uint64_t muladd_small(uint64_t y, uint32_t x, uint64_t cadd) const {
        uint64_t p0 = (uint64_t)(uint32_t)y * x;
        uint64_t p1 = (uint64_t)(uint32_t)(y >> 32) * x + (p0 >> 32);  // < 2^64
        uint64_t lo = (p1 << 32) | (uint32_t)p0;
        uint32_t hi = (uint32_t)(p1 >> 32);                              // y*x = hi:lo
        lo += cadd;
        hi += lo < cadd;                                                 // hi <= 2^32 - 1
        if (K64) {
            uint64_t u = (uint64_t)hi * c + lo;                          // hi*c < 2^63
            if (u < lo) u += c;                                          // wrapped: 2^64 == c
            return csub(u);
        }
        // k < 64: T = xh*2^k + xl,  xh < 2^(96-k) <= 2^63
        uint64_t xh = ((uint64_t)hi << (64 - k)) | (lo >> k);
        uint64_t xl = lo & mask;
        if (C1) {
            uint64_t w = xl + xh;                                        // < 2^64 (k >= 33)
            return csub((w & mask) + (w >> k));
        }
        // xh*c may need 94 bits: fold in two steps
        ff_u128 w = (ff_u128)xh * c + xl;
        uint64_t wh = (uint64_t)(w >> k);
        return csub((ff_lo(w) & mask) + wh * (uint64_t)c);
    }
    FF_HD uint64_t muladd(uint64_t a, uint64_t b, uint64_t cadd) const {
        // a*b + c < p^2 + p < 2^(2k) for k<64; may wrap 128 bits only if k==64
        if (K64) return add(mul(a, b), cadd);
        if (C1) return csub(fold64(presum(a, b) + cadd));
        return red128((ff_u128)a * b + cadd);
    }

    FF_HD void acc_zero(acc& s) const { s.a0 = s.a1 = s.a2 = 0; }
    FF_HD void acc_mac(acc& s, uint64_t lam, uint64_t x) const {
        ff_u128 pr = (ff_u128)lam * x;
        ff_u128 lo = (ff_u128)s.a0 + ff_lo(pr);
        s.a0 = ff_lo(lo);
        ff_u128 mid = (ff_u128)s.a1 + ff_hi(pr) + ff_hi(lo);
        s.a1 = ff_lo(mid);
        s.a2 += ff_hi(mid);
    }
    // value < 2^(2k+8)  (at most 256 products)
    FF_HD uint64_t acc_reduce(const acc& s) const {
        if (K64) {
            ff_u128 hi2 = ff_make128(s.a2, s.a1);    // < 2^72
            ff_u128 t = hi2 * (ff_u128)c + s.a0;     // < 2^104
            return red128(t);
        }
        ff_u128 mid = ff_make128(s.a1, s.a0) >> k;
        ff_u128 top = (ff_u128)s.a2 << (128 - k);
        ff_u128 xh = top | mid;  // < 2^(k+8)
        uint64_t xl = s.a0 & mask;
        ff_u128 t = C1 ? xh + xl : xh * (ff_u128)c + xl;  // < 2^(2k)
        return red128(t);
    }
};

// ---------------------------------------------------------------------------
// RC64: arbitrary modulus 2 <= p < 2^64.  Barrett-type reduction with a
// precomputed 64-bit reciprocal of the normalised modulus (Moeller-Granlund
// "division by invariant integers", 2-by-1 step).  Canonical in/out, so no
// Montgomery domain conversion at the API boundary.
// ---------------------------------------------------------------------------
struct RC64 {
    typedef uint64_t elem;
    typedef uint64_t word;
    enum { EPW = 1 };
    enum { BINARY = 0 };
    uint64_t p;  // modulus
    uint64_t d;  // p << s, top bit set
    uint64_t v;  // floor((2^128-1)/d) - 2^64
    uint32_t s;  // normalisation shift = clz(p)
    uint32_t pad_;

    struct acc {
        uint64_t a0, a1, a2;
    };
    // constants are used as-is (no domain conversion)
    FF_HD uint64_t prep(uint64_t cst) const { return cst; }


    // (u1:u0) mod d, requires u1 < d
    FF_HD uint64_t rem21(uint64_t u1, uint64_t u0) const {
        ff_u128 q = (ff_u128)v * u1 + ff_make128(u1, u0);
        uint64_t q1 = ff_hi(q) + 1;
        uint64_t q0 = ff_lo(q);
        uint64_t r = u0 - q1 * d;
        if (r > q0) r += d;
        if (r >= d) r -= d;
        return r;
    }
    FF_HD uint64_t add(uint64_t a, uint64_t b) const {
        uint64_t t = a + b;
        bool carry = t < a;
        return (carry || t >= p) ? t - p : t;
    }
    FF_HD uint64_t sub(uint64_t a, uint64_t b) const {
        uint64_t t = a - b;
        return a < b ? t + p : t;
    }
    FF_HD uint64_t neg(uint64_t a) const { return a ? p - a : 0; }
    FF_HD uint64_t mul(uint64_t a, uint64_t b) const {
        ff_u128 x = (ff_u128)(a << s) * b;  // (a*b) << s, high limb < d
        return rem21(ff_hi(x), ff_lo(x)) >> s;
    }
    FF_HD uint64_t reduce_raw(uint64_t x) const {
        uint64_t u1 = s ? (x >> (64 - s)) : 0;
        return rem21(u1, x << s) >> s;
    }
    FF_HD uint64_t muladd_small(uint64_t y, uint32_t x, uint64_t cadd) const {
        ff_u128 t = ((ff_u128)y * x + cadd) << s;  // < d * (2^32+1)
        return rem21(ff_hi(t), ff_lo(t)) >> s;
    }
    FF_HD uint64_t muladd(uint64_t a, uint64_t b, uint64_t cadd) const {
        return add(mul(a, b), cadd);
    }
    FF_HD void acc_zero(acc& t) const { t.a0 = t.a1 = t.a2 = 0; }
    FF_HD void acc_mac(acc& t, uint64_t lam, uint64_t x) const {
        ff_u128 pr = (ff_u128)lam * x;
        ff_u128 lo = (ff_u128)t.a0 + ff_lo(pr);
        t.a0 = ff_lo(lo);
        ff_u128 mid = (ff_u128)t.a1 + ff_hi(pr) + ff_hi(lo);
        t.a1 = ff_lo(mid);
        t.a2 += ff_hi(mid);
    }
    FF_HD uint64_t acc_reduce(const acc& t) const {
        // W = V << s as three limbs; V < 2^8 p^2  =>  w2 < 2^8 <= d
        uint64_t w0 = t.a0 << s;
        uint64_t w1 = s ? ((t.a1 << s) | (t.a0 >> (64 - s))) : t.a1;
        uint64_t w2 = s ? ((t.a2 << s) | (t.a1 >> (64 - s))) : t.a2;
        uint64_t r1 = rem21(w2, w1);
        return rem21(r1, w0) >> s;
    }
};

// ---------------------------------------------------------------------------
// RC32: arbitrary modulus 2 <= p < 2^32 stored as uint32 (half the HBM bytes of
// the 64-bit path).  Same reciprocal reduction on 32-bit words.
// ---------------------------------------------------------------------------
struct RC32 {
    typedef uint32_t elem;
    typedef uint32_t word;
    enum { EPW = 1 };
    enum { BINARY = 0 };
    uint32_t p, d, v, s;

    struct acc {
        uint64_t lo;  // sum of 64-bit products, low part
        uint32_t hi;  // carries
    };
    // constants are used as-is (no domain conversion)
    FF_HD uint32_t prep(uint32_t cst) const { return cst; }


    FF_HD uint32_t rem21(uint32_t u1, uint32_t u0) const {
        uint64_t q = (uint64_t)v * u1 + (((uint64_t)u1 << 32) | u0);
        uint32_t q1 = (uint32_t)(q >> 32) + 1;
        uint32_t q0 = (uint32_t)q;
        uint32_t r = u0 - q1 * d;
        if (r > q0) r += d;
        if (r >= d) r -= d;
        return r;
    }
    FF_HD uint32_t add(uint32_t a, uint32_t b) const {
        uint32_t t = a + b;
        bool carry = t < a;
        return (carry || t >= p) ? t - p : t;
    }
    FF_HD uint32_t sub(uint32_t a, uint32_t b) const {
        uint32_t t = a - b;
        return a < b ? t + p : t;
    }
    FF_HD uint32_t neg(uint32_t a) const { return a ? p - a : 0; }
    FF_HD uint32_t mul(uint32_t a, uint32_t b) const {
        uint64_t x = (uint64_t)(a << s) * b;
        return rem21((uint32_t)(x >> 32), (uint32_t)x) >> s;
    }
    FF_HD uint32_t reduce_raw(uint32_t x) const {
        uint32_t u1 = s ? (x >> (32 - s)) : 0;
        return rem21(u1, x << s) >> s;
    }
    // y*x + c with x < 2^32: up to 64 bits + ; two reduction steps
    FF_HD uint32_t muladd_small(uint32_t y, uint32_t x, uint32_t cadd) const {
        // reduce x first (x is a public party index, usually < p already)
        uint32_t xr = x >= p ? reduce_raw(x) : x;
        return add(mul(y, xr), cadd);
    }
    FF_HD uint32_t muladd(uint32_t a, uint32_t b, uint32_t cadd) const { return add(mul(a, b), cadd); }
    FF_HD void acc_zero(acc& t) const {
        t.lo = 0;
        t.hi = 0;
    }
    FF_HD void acc_mac(acc& t, uint32_t lam, uint32_t x) const {
        uint64_t pr = (uint64_t)lam * x;
        uint64_t n = t.lo + pr;
        t.hi += n < pr;
        t.lo = n;
    }
    FF_HD uint32_t acc_reduce(const acc& t) const {
        // V = hi:lo (96 bits, hi < 2^8);  W = V << s
        uint32_t a0 = (uint32_t)t.lo, a1 = (uint32_t)(t.lo >> 32), a2 = t.hi;
        uint32_t w0 = a0 << s;
        uint32_t w1 = s ? ((a1 << s) | (a0 >> (32 - s))) : a1;
        uint32_t w2 = s ? ((a2 << s) | (a1 >> (32 - s))) : a2;
        uint32_t r1 = rem21(w2, w1);
        return rem21(r1, w0) >> s;
    }
};

// ---------------------------------------------------------------------------
// Dot products of multi-limb residues in 28-bit digits (round 6): sum_j lam_j * x_j with every operand cut into NL digits of
// 28 bits and the digit products (< 2^56) added into 2 NL - 1 column sums of 64 bits -- one v_mad_u64_u32 per digit product,
// no carry anywhere until the single conversion at the end.  The 128-bit arithmetic of acc_mac costs the compiler about 40
// instructions per 64 x 64 limb product (carry compares, selects); here a three-limb term is 49 multiply-adds + 10 to cut
// the element, a two-limb term 25 + 7 (the coefficient's digits are wave-uniform: scalar unit).  Recombination of K <= 9
// rows over a 136-bit prime: ~1700 -> ~600 instructions per element, which puts the kernel back on HBM.
// Bound: a column holds at most NL products per term: NL * terms * 2^56 < 2^64 for up to FF_D28_MAX_TERMS terms at NL = 7.
// ---------------------------------------------------------------------------
enum { FF_D28_MAX_TERMS = 32 };
template <int NL>
struct LazyDot {
    uint64_t c[2 * NL - 1];
};
// digit i = bits [28 i, 28 i + 28) of the little-endian 32-bit words w[0..NW)
template <int NW, int NL>
FF_HD void ff_digits28(const uint32_t (&w)[NW], uint32_t (&d)[NL]) {
#pragma unroll
    for (int i = 0; i < NL; ++i) {
        const int bit = 28 * i, q = bit >> 5, off = bit & 31;
        uint32_t v = q < NW ? (w[q < NW ? q : 0] >> off) : 0u;
        if (off > 4 && q + 1 < NW) v |= w[q + 1 < NW ? q + 1 : 0] << (32 - off);
        d[i] = v & 0x0fffffffu;
    }
}
template <int NL>
FF_HD void ff_lazy_zero(LazyDot<NL>& s) {
#pragma unroll
    for (int i = 0; i < 2 * NL - 1; ++i) s.c[i] = 0;
}
template <int NL>
FF_HD void ff_lazy_mac(LazyDot<NL>& s, const uint32_t (&dl)[NL], const uint32_t (&dx)[NL]) {
#pragma unroll
    for (int i = 0; i < NL; ++i)
#pragma unroll
        for (int j = 0; j < NL; ++j) s.c[i + j] += (uint64_t)dx[i] * dl[j];
}
// column sums -> NA little-endian 64-bit limbs.  The value must be below 2^(28 (2 NL - 1) + 28) and below 2^(64 NA).
template <int NL, int NA>
FF_HD void ff_lazy_limbs(const LazyDot<NL>& s, uint64_t (&a)[NA]) {
    constexpr int NC = 2 * NL - 1;
    uint32_t dg[NC + 2];
    uint64_t carry = 0;
#pragma unroll
    for (int i = 0; i < NC; ++i) {
        const uint64_t t = s.c[i] + carry;
        dg[i] = (uint32_t)t & 0x0fffffffu;
        carry = t >> 28;
    }
    dg[NC] = (uint32_t)carry;
    dg[NC + 1] = 0;
#pragma unroll
    for (int j = 0; j < NA; ++j) {
        uint32_t wd[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int bit = 32 * (2 * j + h), q = bit / 28, off = bit - 28 * q;      // off = 0, 4, ..., 24: two digits per word
            const uint32_t lo = q <= NC ? dg[q <= NC ? q : 0] >> off : 0u;
            const uint32_t hi = q + 1 <= NC ? dg[q + 1 <= NC ? q + 1 : 0] << (28 - off) : 0u;
            wd[h] = lo | hi;
        }
        a[j] = (uint64_t)wd[0] | ((uint64_t)wd[1] << 32);
    }
}

// ---------------------------------------------------------------------------
// Product chains in digits (round 6): the exponentiations behind sqrt / inverse sqrt / pow over the multi-limb 2^k - c
// primes.  np_random_bits (runtime.py:4243-4273) takes the inverse square root of f x n opened squares for every
// fixed-point product of n elements -- over the 80-bit default prime of SecFxp() that one kernel was two thirds of the
// product's GPU time.  A value is NL digits of w bits, w = ceil(k / NL) <= 28, so that 2^(w NL) == cs := c 2^(w NL - k)
// (mod p) folds at a DIGIT boundary: a product is NL^2 multiply-adds into column sums (NL (NL + 1) / 2 for a square), one
// carry pass, NL multiply-adds by cs, a second carry pass and one more multiply-add -- ~45 instructions at NL = 3 where
// the 128-bit arithmetic of mul() takes ~100 -- and values stay partially reduced until the chain ends.
// Invariant of every value entering or leaving mul / sqr:  d[i] < 2^w (i < NL - 1),  d[NL - 1] < 2^(w + 1).
// Needs cs < 2^20 (every default prime: c < 2^13); other moduli keep the limb arithmetic.
// ---------------------------------------------------------------------------
template <int NL>
struct DigitChain {
    uint32_t w, mask, cs;
    struct val {
        uint32_t d[NL];
    };
    FF_HD bool setup(uint32_t k, uint32_t c) {
        w = (k + NL - 1) / NL;
        if (w > 28 || w < 22) return false;               // (k >= 65: w >= 22; the bounds below are worked out for 22..28)
        const uint32_t sh = w * NL - k;                      // < NL
        if (c >> (20 - sh)) return false;                     // cs < 2^20, in 32-bit arithmetic throughout (a 64-bit
        cs = c << sh;                                         // intermediate here made every product by cs 64 x 32)
        mask = (1u << w) - 1u;
        return true;
    }
    // column sums (< 2^61) -> value: carry pass, fold the upper NL digits (+ the carry) by cs, carry pass, fold the carry
    FF_HD val reduce(const uint64_t (&col)[2 * NL - 1]) const {
        uint32_t e[2 * NL];
        uint64_t carry = 0;
#pragma unroll
        for (int m = 0; m < 2 * NL - 1; ++m) {
            const uint64_t t = col[m] + carry;
            e[m] = (uint32_t)t & mask;
            carry = t >> w;
        }
        e[2 * NL - 1] = (uint32_t)carry;                     // < 2^(w + 3)
        uint32_t fd[NL];
        carry = 0;
#pragma unroll
        for (int j = 0; j < NL; ++j) {
            const uint64_t t = (uint64_t)e[NL + j] * cs + e[j] + carry;      // < 2^(w + 24)
            fd[j] = (uint32_t)t & mask;
            carry = t >> w;                                   // < 2^24
        }
        val r;
        carry = (uint64_t)(uint32_t)carry * cs + fd[0];       // 2^(w NL) == cs once more: < 2^45
#pragma unroll
        for (int j = 0; j < NL - 1; ++j) {
            r.d[j] = (uint32_t)carry & mask;
            carry = (carry >> w) + fd[j + 1];
        }
        r.d[NL - 1] = (uint32_t)carry;                        // < 2^w + 2^15
        return r;
    }
    FF_HD val mul(const val& x, const val& y) const {
        uint64_t col[2 * NL - 1];
#pragma unroll
        for (int m = 0; m < 2 * NL - 1; ++m) col[m] = 0;
#pragma unroll
        for (int i = 0; i < NL; ++i)
#pragma unroll
            for (int j = 0; j < NL; ++j) col[i + j] += (uint64_t)x.d[i] * y.d[j];
        return reduce(col);
    }
    FF_HD val sqr(const val& x) const {
        uint64_t col[2 * NL - 1];
        uint32_t x2[NL];
#pragma unroll
        for (int i = 0; i < NL; ++i) x2[i] = x.d[i] << 1;     // < 2^30
#pragma unroll
        for (int m = 0; m < 2 * NL - 1; ++m) col[m] = 0;
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            col[2 * i] += (uint64_t)x.d[i] * x.d[i];
#pragma unroll
            for (int j = i + 1; j < NL; ++j) col[i + j] += (uint64_t)x2[i] * x.d[j];
        }
        return reduce(col);
    }
};

// ---------------------------------------------------------------------------
// PM128: prime p = 2^k - c, 65 <= k <= 128, c < 2^31, two 64-bit limbs
// (2^128-173, 2^127-1, 2^96-17, the 80-bit SecFxp default, ...).
// ---------------------------------------------------------------------------
template <bool K128>
struct PM128 {
    typedef u128e elem;
    typedef u128e word;
    enum { EPW = 1 };
    enum { BINARY = 0 };
    uint64_t p_lo, p_hi;        // modulus
    uint64_t mask_lo, mask_hi;  // 2^k - 1
    uint32_t c;                 // 2^k - p
    uint32_t k;

    struct acc {
        uint64_t a0, a1, a2, a3, a4;
    };
    // constants are used as-is (no domain conversion)
    FF_HD u128e prep(u128e cst) const { return cst; }


    FF_HD ff_u128 P() const { return ff_make128(p_hi, p_lo); }
    FF_HD ff_u128 M() const { return ff_make128(mask_hi, mask_lo); }
    static FF_HD ff_u128 U(const u128e& a) { return ff_make128(a.hi, a.lo); }
    static FF_HD u128e E(ff_u128 x) {
        u128e r;
        r.lo = ff_lo(x);
        r.hi = ff_hi(x);
        return r;
    }
    FF_HD ff_u128 csub(ff_u128 x) const { return x >= P() ? x - P() : x; }

    FF_HD ff_u128 addu(ff_u128 a, ff_u128 b) const {
        ff_u128 t = a + b;
        if (K128) {
            bool carry = t < a;
            return (carry || t >= P()) ? t - P() : t;
        }
        return csub(t);
    }
    FF_HD u128e add(const u128e& a, const u128e& b) const { return E(addu(U(a), U(b))); }
    FF_HD u128e sub(const u128e& a, const u128e& b) const {
        ff_u128 x = U(a), y = U(b);
        ff_u128 t = x - y;
        return E(x < y ? t + P() : t);
    }
    FF_HD u128e neg(const u128e& a) const {
        ff_u128 x = U(a);
        return E(x ? P() - x : 0);
    }

    // Final stage: value T = t2*2^128 + (t1:t0) < 2^(k+64)  ->  canonical.
    template <bool TWO_FOLDS>
    FF_HD ff_u128 fold3(uint64_t t2, ff_u128 t10) const {
        ff_u128 u;
        if (K128) {
            ff_u128 add_ = (ff_u128)t2 * c;
            u = t10 + add_;
            if (u < add_) u += c;  // wrapped 2^128 == c
            return csub(u);
        }
        // wh = T >> k  (fits 64 bits), wl = T & mask
        uint32_t sh = k - 64;  // 1..63
        uint64_t wh = (ff_hi(t10) >> sh) | (t2 << (64 - sh));
        ff_u128 wl = t10 & M();
        u = wl + (ff_u128)wh * c;
        if (TWO_FOLDS) {
            uint64_t wh2 = ff_hi(u) >> sh;
            u = (u & M()) + (ff_u128)wh2 * c;
        }
        return csub(u);
    }

    // 256-bit product as four limbs
    static FF_HD void mul256(ff_u128 a, ff_u128 b, uint64_t x[4]) {
        uint64_t a0 = ff_lo(a), a1 = ff_hi(a), b0 = ff_lo(b), b1 = ff_hi(b);
        ff_u128 p00 = (ff_u128)a0 * b0;
        ff_u128 p01 = (ff_u128)a0 * b1;
        ff_u128 p10 = (ff_u128)a1 * b0;
        ff_u128 p11 = (ff_u128)a1 * b1;
        x[0] = ff_lo(p00);
        ff_u128 m = (ff_u128)ff_hi(p00) + ff_lo(p01) + ff_lo(p10);
        x[1] = ff_lo(m);
        ff_u128 h = (ff_u128)ff_hi(m) + ff_hi(p01) + ff_hi(p10) + ff_lo(p11);
        x[2] = ff_lo(h);
        x[3] = ff_hi(p11) + ff_hi(h);
    }

    // value (x3:x2:x1:x0) < 2^(2k)
    FF_HD ff_u128 red256(const uint64_t x[4]) const {
        ff_u128 xh, xl;
        if (K128) {
            xh = ff_make128(x[3], x[2]);
            xl = ff_make128(x[1], x[0]);
        } else {
            uint32_t sh = k - 64;  // 1..63
            uint64_t h0 = (x[1] >> sh) | (x[2] << (64 - sh));
            uint64_t h1 = (x[2] >> sh) | (x[3] << (64 - sh));
            xh = ff_make128(h1, h0);
            xl = ff_make128(x[1], x[0]) & M();
        }
        // T = xh*c + xl  as (t2 : t10)
        ff_u128 pl = (ff_u128)ff_lo(xh) * c;
        ff_u128 ph = (ff_u128)ff_hi(xh) * c;
        ff_u128 s0 = xl + pl;
        uint64_t t2 = (s0 < pl) ? 1 : 0;
        ff_u128 phs = ph << 64;
        ff_u128 s1 = s0 + phs;
        t2 += (s1 < phs) ? 1 : 0;
        t2 += ff_hi(ph);
        return fold3<false>(t2, s1);
    }

    FF_HD u128e mul(const u128e& a, const u128e& b) const {
        uint64_t x[4];
        mul256(U(a), U(b), x);
        return E(red256(x));
    }
    FF_HD u128e reduce_raw(const u128e& a) const {
        if (K128) return E(csub(U(a)));
        return E(fold3<false>(0, U(a)));
    }
    FF_HD u128e muladd_small(const u128e& y, uint32_t x, const u128e& cadd) const {
        // V = y*x + c : three limbs
        ff_u128 l = (ff_u128)y.lo * x;
        ff_u128 h = (ff_u128)y.hi * x + ff_hi(l);
        ff_u128 v10 = ff_make128(ff_lo(h), ff_lo(l));
        uint64_t v2 = ff_hi(h);
        ff_u128 cc = U(cadd);
        ff_u128 t = v10 + cc;
        if (t < cc) v2 += 1;
        return E(fold3<false>(v2, t));
    }
    FF_HD u128e muladd(const u128e& a, const u128e& b, const u128e& cadd) const {
        return add(mul(a, b), cadd);
    }


    // product chains in digits (see DigitChain): NL = ceil(k / 28) digits -- 3 up to 84 bits, 4 up to 112, 5 beyond
    enum { CHAIN_MIN_NL = 3, CHAIN_MAX_NL = 5 };
    template <int NL>
    FF_HD bool chain_setup(DigitChain<NL>& dc) const { return dc.setup(k, c); }
    template <int NL>
    FF_HD typename DigitChain<NL>::val chain_in(const DigitChain<NL>& dc, const u128e& a) const {
        const ff_u128 v = U(a);
        typename DigitChain<NL>::val r;
#pragma unroll
        for (int i = 0; i < NL; ++i) r.d[i] = (uint32_t)(v >> (dc.w * i)) & dc.mask;
        return r;
    }
    template <int NL>
    FF_HD u128e chain_out(const DigitChain<NL>& dc, const typename DigitChain<NL>::val& x) const {
        uint64_t t[3] = {0, 0, 0};                           // the digits do not overlap: OR them into place
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            const uint32_t pos = dc.w * i, q = pos >> 6, sh = pos & 63;
            const uint64_t lo = (uint64_t)x.d[i] << sh, hi = sh ? (uint64_t)x.d[i] >> (64 - sh) : 0;
            if (q == 0) { t[0] |= lo; t[1] |= hi; } else { t[1] |= lo; t[2] |= hi; }
        }
        return E(fold3<true>(t[2], ff_make128(t[1], t[0])));  // < 2^(k + 5) -> canonical
    }

    // dot products of at most FF_D28_MAX_TERMS terms in 28-bit digits (see LazyDot): five digits cover 128 bits
    typedef LazyDot<5> lacc;
    FF_HD void lacc_zero(lacc& s) const { ff_lazy_zero(s); }
    FF_HD void lacc_mac(lacc& s, const u128e& lam, const u128e& xe) const {
        const uint32_t wl[4] = {(uint32_t)lam.lo, (uint32_t)(lam.lo >> 32), (uint32_t)lam.hi, (uint32_t)(lam.hi >> 32)};
        const uint32_t wx[4] = {(uint32_t)xe.lo, (uint32_t)(xe.lo >> 32), (uint32_t)xe.hi, (uint32_t)(xe.hi >> 32)};
        uint32_t dl[5], dx[5];
        ff_digits28(wl, dl);
        ff_digits28(wx, dx);
        ff_lazy_mac(s, dl, dx);
    }
    // the same with the operands already cut into digits (kernels that use an operand many times: the skinny products)
    enum { LAZY_NL = 5 };
    FF_HD void lacc_digits(const u128e& x, uint32_t (&d)[5]) const {
        const uint32_t w[4] = {(uint32_t)x.lo, (uint32_t)(x.lo >> 32), (uint32_t)x.hi, (uint32_t)(x.hi >> 32)};
        ff_digits28(w, d);
    }
    FF_HD void lacc_mac_digits(lacc& s, const uint32_t (&dl)[5], const uint32_t (&dx)[5]) const { ff_lazy_mac(s, dl, dx); }
    FF_HD u128e lacc_reduce(const lacc& s) const {
        uint64_t a[5];
        ff_lazy_limbs(s, a);
        acc t;
        t.a0 = a[0]; t.a1 = a[1]; t.a2 = a[2]; t.a3 = a[3]; t.a4 = a[4];
        return acc_reduce(t);
    }

    FF_HD void acc_zero(acc& s) const { s.a0 = s.a1 = s.a2 = s.a3 = s.a4 = 0; }
    FF_HD void acc_mac(acc& s, const u128e& lam, const u128e& xe) const {
        uint64_t x[4];
        mul256(U(lam), U(xe), x);
        ff_u128 t = (ff_u128)s.a0 + x[0];
        s.a0 = ff_lo(t);
        t = (ff_u128)s.a1 + x[1] + ff_hi(t);
        s.a1 = ff_lo(t);
        t = (ff_u128)s.a2 + x[2] + ff_hi(t);
        s.a2 = ff_lo(t);
        t = (ff_u128)s.a3 + x[3] + ff_hi(t);
        s.a3 = ff_lo(t);
        s.a4 += ff_hi(t);
    }
    // V < 2^(2k+8)
    FF_HD u128e acc_reduce(const acc& s) const {
        // xh = V >> k (three limbs h2:h1:h0, < 2^(k+8)), xl = V & mask
        uint64_t h0, h1, h2;
        ff_u128 xl;
        if (K128) {
            h0 = s.a2;
            h1 = s.a3;
            h2 = s.a4;
            xl = ff_make128(s.a1, s.a0);
        } else {
            uint32_t sh = k - 64;
            h0 = (s.a1 >> sh) | (s.a2 << (64 - sh));
            h1 = (s.a2 >> sh) | (s.a3 << (64 - sh));
            h2 = (s.a3 >> sh) | (s.a4 << (64 - sh));
            xl = ff_make128(s.a1, s.a0) & M();
        }
        // T = xh*c + xl  (< 2^(k+40)): limbs t2 : t10
        ff_u128 q0 = (ff_u128)h0 * c;
        ff_u128 q1 = (ff_u128)h1 * c + ff_hi(q0);
        ff_u128 q2 = (ff_u128)h2 * c + ff_hi(q1);
        ff_u128 prod10 = ff_make128(ff_lo(q1), ff_lo(q0));
        uint64_t t2 = ff_lo(q2);
        ff_u128 t10 = prod10 + xl;
        if (t10 < xl) t2 += 1;
        if (K128) {
            // t2 < 2^40: (t1:t0) + t2*c, one wrap possible
            return E(fold3<false>(t2, t10));
        }
        return E(fold3<true>(t2, t10));
    }
};

// ---------------------------------------------------------------------------
// PM96: the PM128 arithmetic for 65 <= k <= 96 with elements STORED in 12 bytes
// (three little-endian 32-bit limbs = the reference's field.to_bytes layout for these
// fields, finfields.py:91-102).  A lane moves one element per global_load/store_dwordx3,
// which streams as fast as dwordx4 (tools/tune_x3.hip: 6.5 TB/s), so every HBM-bound
// kernel gains the 25 % the narrower storage saves.  Default SecInt(33..64) fields land here.
// ---------------------------------------------------------------------------
struct e96 {
    uint32_t x[3];
};
struct PM96 : PM128<false> {
    typedef e96 elem;
    enum { CHAIN_MIN_NL = 3, CHAIN_MAX_NL = 4 };          // (at most 96 bits)
    // residues below 2^96: four 28-bit digits (16 digit products per term instead of 25)
    typedef LazyDot<4> lacc;
    FF_HD void lacc_zero(lacc& s) const { ff_lazy_zero(s); }
    FF_HD void lacc_mac(lacc& s, const u128e& lam, const u128e& xe) const {
        const uint32_t wl[3] = {(uint32_t)lam.lo, (uint32_t)(lam.lo >> 32), (uint32_t)lam.hi};
        const uint32_t wx[3] = {(uint32_t)xe.lo, (uint32_t)(xe.lo >> 32), (uint32_t)xe.hi};
        uint32_t dl[4], dx[4];
        ff_digits28(wl, dl);
        ff_digits28(wx, dx);
        ff_lazy_mac(s, dl, dx);
    }
    enum { LAZY_NL = 4 };
    FF_HD void lacc_digits(const u128e& x, uint32_t (&d)[4]) const {
        const uint32_t w[3] = {(uint32_t)x.lo, (uint32_t)(x.lo >> 32), (uint32_t)x.hi};
        ff_digits28(w, d);
    }
    FF_HD void lacc_mac_digits(lacc& s, const uint32_t (&dl)[4], const uint32_t (&dx)[4]) const { ff_lazy_mac(s, dl, dx); }
    FF_HD u128e lacc_reduce(const lacc& s) const {
        uint64_t a[4];
        ff_lazy_limbs(s, a);                              // < 2^(192 + 5)
        acc t;
        t.a0 = a[0]; t.a1 = a[1]; t.a2 = a[2]; t.a3 = a[3]; t.a4 = 0;
        return acc_reduce(t);
    }
};

// ---------------------------------------------------------------------------
// PM192: prime p = 2^k - c, 129 <= k <= 192, c < 2^31, three 64-bit limbs, 24-byte storage
// (SecInt(97..160): the l + 32-bit default fields, e.g. the 136-bit field of the largest
// demos/np_lpsolver.py dataset).  Same fold-with-2^k == c reduction as PM128, one limb wider:
// a product is 9 limb products + a 3-limb fold; every kernel stays HBM-bound at 24 bytes per
// element.  Share generation uses the Horner step (no lazy accumulator).
// ---------------------------------------------------------------------------
struct u192e {
    uint64_t lo, mid, hi;
};

// cond ? a : b on words.  For the three-limb struct the choice is made limb by limb: `cond ? x : y` on two
// struct lvalues becomes a choice between their ADDRESSES, which keeps local arrays of them in scratch memory.
template <class W>
FF_HD W ff_pick(bool cond, const W& a, const W& b) {
    return cond ? a : b;
}
FF_HD u192e ff_pick(bool cond, const u192e& a, const u192e& b) {
    u192e r;
    r.lo = cond ? a.lo : b.lo;
    r.mid = cond ? a.mid : b.mid;
    r.hi = cond ? a.hi : b.hi;
    return r;
}

struct PM192 {
    typedef u192e elem;
    typedef u192e word;
    enum { EPW = 1 };
    enum { BINARY = 0 };
    uint64_t p0, p1, p2;   // modulus
    uint64_t mask_hi;      // top limb of 2^k - 1 (the two low limbs are all ones)
    uint32_t c;            // 2^k - p
    uint32_t k;

    struct acc {
        uint64_t a[7];     // unreduced dot product: < 2^(2k + 8)
    };
    FF_HD u192e prep(u192e cst) const { return cst; }

    static FF_HD bool ge(const u192e& a, const u192e& b) {
        if (a.hi != b.hi) return a.hi > b.hi;
        if (a.mid != b.mid) return a.mid > b.mid;
        return a.lo >= b.lo;
    }
    static FF_HD u192e add3(const u192e& a, const u192e& b, uint64_t& carry) {
        u192e r;
        ff_u128 t = (ff_u128)a.lo + b.lo;
        r.lo = ff_lo(t);
        t = (ff_u128)a.mid + b.mid + ff_hi(t);
        r.mid = ff_lo(t);
        t = (ff_u128)a.hi + b.hi + ff_hi(t);
        r.hi = ff_lo(t);
        carry = ff_hi(t);
        return r;
    }
    static FF_HD u192e sub3(const u192e& a, const u192e& b) {   // a - b mod 2^192
        u192e r;
        r.lo = a.lo - b.lo;
        uint64_t br = a.lo < b.lo;
        uint64_t m = a.mid - b.mid;
        uint64_t br2 = (a.mid < b.mid) | ((m < br) ? 1u : 0u);
        r.mid = m - br;
        r.hi = a.hi - b.hi - br2;
        return r;
    }
    FF_HD u192e P() const {
        u192e r;
        r.lo = p0;
        r.mid = p1;
        r.hi = p2;
        return r;
    }
    FF_HD u192e csub(const u192e& x) const { return ge(x, P()) ? sub3(x, P()) : x; }
    FF_HD u192e add(const u192e& a, const u192e& b) const {
        uint64_t carry;
        u192e t = add3(a, b, carry);
        return (carry || ge(t, P())) ? sub3(t, P()) : t;      // carry only when k = 192
    }
    FF_HD u192e sub(const u192e& a, const u192e& b) const {
        u192e t = sub3(a, b);
        if (!ge(a, b)) {
            uint64_t carry;
            t = add3(t, P(), carry);
        }
        return t;
    }
    FF_HD u192e neg(const u192e& a) const {
        if ((a.lo | a.mid | a.hi) == 0) return a;
        return sub3(P(), a);
    }
    // 64-bit word j of (X >> k) for a little-endian limb array X of n limbs (zero beyond).  129 <= k <= 192, so the
    // word sits at limb 2 + j (shifted by k - 128) or, for k = 192, at limb 3 + j: constant indices once j is
    // unrolled -- a runtime limb index would put the array in scratch memory.
    FF_HD uint64_t word_above_k(const uint64_t* x, int n, int j) const {
        const int s = (int)(k & 63);
        const uint64_t w0 = 2 + j < n ? x[2 + j] : 0, w1 = 3 + j < n ? x[3 + j] : 0;
        return k == 192 ? w1 : (w0 >> s) | (w1 << ((64 - s) & 63));
    }
    // T = (t3 : t2 : t1 : t0) < 2^(k + 63)  ->  canonical.  One fold leaves < 2^k + 2^94; TWO_FOLDS handles what
    // sticks out after the first (needed when T >= 2^(k + 32)).
    template <bool TWO_FOLDS>
    FF_HD u192e fold(const uint64_t t[4]) const {
        uint64_t wh = word_above_k(t, 4, 0);
        u192e u;
        u.lo = t[0];
        u.mid = t[1];
        u.hi = t[2] & mask_hi;
        uint64_t over = 0;                                 // bit 192 of the running sum (k = 192 only)
#pragma unroll
        for (int pass = 0; pass < (TWO_FOLDS ? 2 : 1); ++pass) {
            const ff_u128 add_ = (ff_u128)wh * c;
            ff_u128 s = (ff_u128)u.lo + ff_lo(add_);
            u.lo = ff_lo(s);
            s = (ff_u128)u.mid + ff_hi(add_) + ff_hi(s);
            u.mid = ff_lo(s);
            s = (ff_u128)u.hi + ff_hi(s);
            u.hi = ff_lo(s);
            over += ff_hi(s);
            if (TWO_FOLDS && pass == 0) {
                const uint64_t lim[4] = {u.lo, u.mid, u.hi, over};
                wh = word_above_k(lim, 4, 0);
                u.hi &= mask_hi;
                over = 0;
            }
        }
        if (over) {                                        // k = 192: 2^192 == c
            uint64_t carry;
            u192e cc;
            cc.lo = c;
            cc.mid = cc.hi = 0;
            u = add3(u, cc, carry);
        }
        return csub(csub(u));
    }
    static FF_HD void mul384(const u192e& a, const u192e& b, uint64_t x[6]) {
        const uint64_t al[3] = {a.lo, a.mid, a.hi}, bl[3] = {b.lo, b.mid, b.hi};
        ff_u128 col = 0;        // running column sum (low 128 bits) ...
        uint64_t top = 0;       // ... and its overflow into bit 128+
#pragma unroll
        for (int kk = 0; kk < 5; ++kk) {
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const int j = kk - i;
                if (j < 0 || j > 2) continue;
                const ff_u128 pr = (ff_u128)al[i] * bl[j];
                col += pr;
                top += col < pr ? 1 : 0;
            }
            x[kk] = ff_lo(col);
            col = (col >> 64) | ((ff_u128)top << 64);
            top = 0;
        }
        x[5] = ff_lo(col);
    }
    // (x5..x0) < 2^(2k) -> canonical:  T = (X >> k) * c + (X & mask)  < 2^(k + 32)
    FF_HD u192e red384(const uint64_t x[6]) const {
        uint64_t h[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) h[j] = word_above_k(x, 6, j);
        uint64_t t[4];
        ff_u128 s = (ff_u128)h[0] * c + x[0];
        t[0] = ff_lo(s);
        s = (ff_u128)h[1] * c + x[1] + ff_hi(s);
        t[1] = ff_lo(s);
        s = (ff_u128)h[2] * c + (x[2] & mask_hi) + ff_hi(s);
        t[2] = ff_lo(s);
        t[3] = ff_hi(s);
        return fold<false>(t);
    }
    FF_HD u192e mul(const u192e& a, const u192e& b) const {
        uint64_t x[6];
        mul384(a, b, x);
        return red384(x);
    }
    FF_HD u192e reduce_raw(const u192e& a) const {
        const uint64_t t[4] = {a.lo, a.mid, a.hi, 0};
        return fold<false>(t);
    }
    FF_HD u192e muladd_small(const u192e& y, uint32_t x, const u192e& cadd) const {
        uint64_t t[4];
        ff_u128 s = (ff_u128)y.lo * x + cadd.lo;
        t[0] = ff_lo(s);
        s = (ff_u128)y.mid * x + cadd.mid + ff_hi(s);
        t[1] = ff_lo(s);
        s = (ff_u128)y.hi * x + cadd.hi + ff_hi(s);
        t[2] = ff_lo(s);
        t[3] = ff_hi(s);                                   // V < 2^(k + 33)
        return fold<true>(t);
    }
    FF_HD u192e muladd(const u192e& a, const u192e& b, const u192e& cadd) const { return add(mul(a, b), cadd); }


    // product chains in digits (see DigitChain): NL = ceil(k / 28) digits -- 5 up to 140 bits, 6 up to 168, 7 beyond
    enum { CHAIN_MIN_NL = 5, CHAIN_MAX_NL = 7 };
    template <int NL>
    FF_HD bool chain_setup(DigitChain<NL>& dc) const { return dc.setup(k, c); }
    template <int NL>
    FF_HD typename DigitChain<NL>::val chain_in(const DigitChain<NL>& dc, const u192e& a) const {
        typename DigitChain<NL>::val r;
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            const uint32_t pos = dc.w * i, q = pos >> 6, sh = pos & 63;     // wave-uniform: selects, not indexing
            const uint64_t lo = q == 0 ? a.lo : q == 1 ? a.mid : a.hi;
            const uint64_t hi = q == 0 ? a.mid : q == 1 ? a.hi : 0;
            const uint64_t v = sh ? (lo >> sh) | (hi << (64 - sh)) : lo;
            r.d[i] = (uint32_t)v & dc.mask;
        }
        return r;
    }
    template <int NL>
    FF_HD u192e chain_out(const DigitChain<NL>& dc, const typename DigitChain<NL>::val& x) const {
        uint64_t t0 = 0, t1 = 0, t2 = 0, t3 = 0;             // the digits do not overlap: OR them into place
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            const uint32_t pos = dc.w * i, q = pos >> 6, sh = pos & 63;
            const uint64_t lo = (uint64_t)x.d[i] << sh, hi = sh ? (uint64_t)x.d[i] >> (64 - sh) : 0;
            if (q == 0) { t0 |= lo; t1 |= hi; } else if (q == 1) { t1 |= lo; t2 |= hi; } else { t2 |= lo; t3 |= hi; }
        }
        const uint64_t t[4] = {t0, t1, t2, t3};
        return fold<false>(t);                               // < 2^(k + 8) -> canonical
    }

    // dot products of at most FF_D28_MAX_TERMS terms in 28-bit digits (see LazyDot): seven digits cover 192 bits
    typedef LazyDot<7> lacc;
    FF_HD void lacc_zero(lacc& s) const { ff_lazy_zero(s); }
    FF_HD void lacc_mac(lacc& s, const u192e& lam, const u192e& xe) const {
        const uint32_t wl[6] = {(uint32_t)lam.lo, (uint32_t)(lam.lo >> 32), (uint32_t)lam.mid, (uint32_t)(lam.mid >> 32),
                                (uint32_t)lam.hi, (uint32_t)(lam.hi >> 32)};
        const uint32_t wx[6] = {(uint32_t)xe.lo, (uint32_t)(xe.lo >> 32), (uint32_t)xe.mid, (uint32_t)(xe.mid >> 32),
                                (uint32_t)xe.hi, (uint32_t)(xe.hi >> 32)};
        uint32_t dl[7], dx[7];
        ff_digits28(wl, dl);
        ff_digits28(wx, dx);
        ff_lazy_mac(s, dl, dx);
    }
    enum { LAZY_NL = 7 };
    FF_HD void lacc_digits(const u192e& x, uint32_t (&d)[7]) const {
        const uint32_t w[6] = {(uint32_t)x.lo, (uint32_t)(x.lo >> 32), (uint32_t)x.mid, (uint32_t)(x.mid >> 32),
                               (uint32_t)x.hi, (uint32_t)(x.hi >> 32)};
        ff_digits28(w, d);
    }
    FF_HD void lacc_mac_digits(lacc& s, const uint32_t (&dl)[7], const uint32_t (&dx)[7]) const { ff_lazy_mac(s, dl, dx); }
    FF_HD u192e lacc_reduce(const lacc& s) const {
        acc t;
        ff_lazy_limbs(s, t.a);
        return acc_reduce(t);
    }

    FF_HD void acc_zero(acc& s) const {
#pragma unroll
        for (int i = 0; i < 7; ++i) s.a[i] = 0;
    }
    FF_HD void acc_mac(acc& s, const u192e& lam, const u192e& xe) const {
        uint64_t x[6];
        mul384(lam, xe, x);
        uint64_t carry = 0;
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const ff_u128 t = (ff_u128)s.a[i] + x[i] + carry;
            s.a[i] = ff_lo(t);
            carry = ff_hi(t);
        }
        s.a[6] += carry;
    }
    // V < 2^(2k + 8): xh = V >> k (< 2^(k+8): four limbs), T = xh * c + (V & mask) < 2^(k + 40)
    FF_HD u192e acc_reduce(const acc& s) const {
        uint64_t h[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) h[j] = word_above_k(s.a, 7, j);
        uint64_t t[4];
        ff_u128 v = (ff_u128)h[0] * c + s.a[0];
        t[0] = ff_lo(v);
        v = (ff_u128)h[1] * c + s.a[1] + ff_hi(v);
        t[1] = ff_lo(v);
        v = (ff_u128)h[2] * c + (s.a[2] & mask_hi) + ff_hi(v);
        t[2] = ff_lo(v);
        // h[3] < 2^8 contributes h[3] * c * 2^192 = h[3] * c * (2^(192-k) mod-free shift): fold it as limb 3
        v = (ff_u128)h[3] * c + ff_hi(v);
        t[3] = ff_lo(v);                                   // < 2^40
        return fold<true>(t);
    }
};

// ---------------------------------------------------------------------------
// MONT128: arbitrary odd modulus 2^64 < p < 2^128 (two limbs).  Canonical
// in/out: mul(a,b) = REDC(REDC(a*b) * R^2) with R = 2^128, i.e. two word-serial
// Montgomery reductions; chains (Horner, dot products) stay cheap because the
// public operand is pre-scaled on the host where possible.
// ---------------------------------------------------------------------------
struct MONT128 {
    typedef u128e elem;
    typedef u128e word;
    enum { EPW = 1 };
    enum { BINARY = 0 };
    uint64_t p_lo, p_hi;
    uint64_t r2_lo, r2_hi;  // R^2 mod p
    uint64_t pinv;          // -p^{-1} mod 2^64
    uint64_t pad_;

    struct acc {
        uint64_t a0, a1;  // running canonical sum (reduced every step)
    };
    // Lagrange / scalar constants are pre-scaled by R so that one REDC per
    // product suffices in acc_mac (defined below, after montmul)


    FF_HD ff_u128 P() const { return ff_make128(p_hi, p_lo); }
    static FF_HD ff_u128 U(const u128e& a) { return ff_make128(a.hi, a.lo); }
    static FF_HD u128e E(ff_u128 x) {
        u128e r;
        r.lo = ff_lo(x);
        r.hi = ff_hi(x);
        return r;
    }
    FF_HD ff_u128 addu(ff_u128 a, ff_u128 b) const {
        ff_u128 t = a + b;
        bool carry = t < a;
        return (carry || t >= P()) ? t - P() : t;
    }
    FF_HD ff_u128 subu(ff_u128 a, ff_u128 b) const {
        ff_u128 t = a - b;
        return a < b ? t + P() : t;
    }
    FF_HD u128e add(const u128e& a, const u128e& b) const { return E(addu(U(a), U(b))); }
    FF_HD u128e sub(const u128e& a, const u128e& b) const { return E(subu(U(a), U(b))); }
    FF_HD u128e neg(const u128e& a) const {
        ff_u128 x = U(a);
        return E(x ? P() - x : 0);
    }
    // REDC of T = (x3:x2:x1:x0) < p * 2^128  ->  T * 2^-128 mod p, canonical
    FF_HD ff_u128 redc(const uint64_t x[4]) const {
        uint64_t t0 = x[0], t1 = x[1], t2 = x[2], t3 = x[3], t4 = 0;
        for (int i = 0; i < 2; ++i) {
            uint64_t m = t0 * pinv;
            ff_u128 c0 = (ff_u128)m * p_lo + t0;  // low limb becomes 0
            ff_u128 c1 = (ff_u128)m * p_hi + t1 + ff_hi(c0);
            ff_u128 c2 = (ff_u128)t2 + ff_hi(c1);
            ff_u128 c3 = (ff_u128)t3 + ff_hi(c2);
            t0 = ff_lo(c1);
            t1 = ff_lo(c2);
            t2 = ff_lo(c3);
            t3 = t4 + ff_hi(c3);
            t4 = 0;
        }
        // result = (t2:t1:t0) < 2p, t2 in {0,1}
        ff_u128 r = ff_make128(t1, t0);
        if (t2 || r >= P()) r -= P();
        return r;
    }
    FF_HD ff_u128 montmul(ff_u128 a, ff_u128 b) const {
        uint64_t x[4];
        PM128<true>::mul256(a, b, x);
        return redc(x);
    }
    FF_HD u128e prep(const u128e& cst) const { return E(montmul(U(cst), ff_make128(r2_hi, r2_lo))); }
    FF_HD u128e mul(const u128e& a, const u128e& b) const {
        ff_u128 t = montmul(U(a), U(b));                   // a*b/R
        return E(montmul(t, ff_make128(r2_hi, r2_lo)));     // * R^2 / R = a*b
    }
    FF_HD u128e reduce_raw(const u128e& a) const {
        // p > 2^64 so at most 2^64 subtractions would be needed in the worst
        // case; use Montgomery instead: a * R^2 / R / ... -> a*1: REDC(a*R2) = a*R,
        // REDC(a*R) = a.  Both products are < p*2^128 only if a < 2^128: true.
        // a*R2 < 2^128 * p ok.
        ff_u128 t = montmul(U(a), ff_make128(r2_hi, r2_lo));  // a*R mod p
        uint64_t x[4] = {ff_lo(t), ff_hi(t), 0, 0};
        return E(redc(x));
    }
    FF_HD u128e muladd_small(const u128e& y, uint32_t x, const u128e& cadd) const {
        u128e xe;
        xe.lo = x;
        xe.hi = 0;
        return add(mul(y, xe), cadd);
    }
    FF_HD u128e muladd(const u128e& a, const u128e& b, const u128e& cadd) const {
        return add(mul(a, b), cadd);
    }
    FF_HD void acc_zero(acc& s) const { s.a0 = s.a1 = 0; }
    // host pre-scales lambda by R (lam' = lam*R mod p) so one REDC per term suffices
    FF_HD void acc_mac(acc& s, const u128e& lamR, const u128e& xe) const {
        ff_u128 t = montmul(U(lamR), U(xe));
        ff_u128 r = addu(ff_make128(s.a1, s.a0), t);
        s.a0 = ff_lo(r);
        s.a1 = ff_hi(r);
    }
    FF_HD u128e acc_reduce(const acc& s) const {
        u128e r;
        r.lo = s.a0;
        r.hi = s.a1;
        return r;
    }
};

// ---------------------------------------------------------------------------
// MONT192: arbitrary odd primes of 129..192 bits (e.g. the "root of unity" primes 1 + 2n(3 + 2j) that
// sectypes.SecInt(l, n=N) asks find_prime_root for, finfields.py:332-343), three 64-bit limbs, 24-byte storage.
// Montgomery arithmetic with R = 2^192 INSIDE a product only: values in memory are canonical residues
// (finfields.py:66,708), as for MONT128.
// ---------------------------------------------------------------------------
struct MONT192 {
    typedef u192e elem;
    typedef u192e word;
    enum { EPW = 1 };
    enum { BINARY = 0 };
    uint64_t p0, p1, p2;        // modulus
    uint64_t r2_0, r2_1, r2_2;  // R^2 mod p
    uint64_t pinv;              // -p^{-1} mod 2^64
    uint64_t pad_;

    struct acc {
        u192e v;                // running canonical sum
    };
    FF_HD u192e P() const {
        u192e r;
        r.lo = p0;
        r.mid = p1;
        r.hi = p2;
        return r;
    }
    FF_HD u192e R2() const {
        u192e r;
        r.lo = r2_0;
        r.mid = r2_1;
        r.hi = r2_2;
        return r;
    }
    FF_HD u192e csub(const u192e& x) const { return PM192::ge(x, P()) ? PM192::sub3(x, P()) : x; }
    FF_HD u192e add(const u192e& a, const u192e& b) const {
        uint64_t carry;
        const u192e t = PM192::add3(a, b, carry);
        return (carry || PM192::ge(t, P())) ? PM192::sub3(t, P()) : t;
    }
    FF_HD u192e sub(const u192e& a, const u192e& b) const {
        u192e t = PM192::sub3(a, b);
        if (!PM192::ge(a, b)) {
            uint64_t carry;
            t = PM192::add3(t, P(), carry);
        }
        return t;
    }
    FF_HD u192e neg(const u192e& a) const {
        if ((a.lo | a.mid | a.hi) == 0) return a;
        return PM192::sub3(P(), a);
    }
    // REDC of T = (x5..x0) < p * 2^192  ->  T * 2^-192 mod p, canonical (word-serial, three rounds)
    FF_HD u192e redc(const uint64_t x[6]) const {
        uint64_t t0 = x[0], t1 = x[1], t2 = x[2], t3 = x[3], t4 = x[4], t5 = x[5], t6 = 0;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const uint64_t m = t0 * pinv;
            ff_u128 c = (ff_u128)m * p0 + t0;              // low limb becomes 0
            c = (ff_u128)m * p1 + t1 + ff_hi(c);
            const uint64_t n0 = ff_lo(c);
            c = (ff_u128)m * p2 + t2 + ff_hi(c);
            const uint64_t n1 = ff_lo(c);
            c = (ff_u128)t3 + ff_hi(c);
            const uint64_t n2 = ff_lo(c);
            c = (ff_u128)t4 + ff_hi(c);
            const uint64_t n3 = ff_lo(c);
            c = (ff_u128)t5 + ff_hi(c);
            const uint64_t n4 = ff_lo(c);
            const uint64_t n5 = t6 + ff_hi(c);
            t0 = n0; t1 = n1; t2 = n2; t3 = n3; t4 = n4; t5 = n5; t6 = 0;
        }
        // result = (t3 : t2 : t1 : t0) < 2p, t3 in {0, 1}
        u192e r;
        r.lo = t0;
        r.mid = t1;
        r.hi = t2;
        return (t3 || PM192::ge(r, P())) ? PM192::sub3(r, P()) : r;
    }
    FF_HD u192e montmul(const u192e& a, const u192e& b) const {
        uint64_t x[6];
        PM192::mul384(a, b, x);
        return redc(x);
    }
    // constants of dot products / gates are pre-scaled by R: one REDC per term (acc_mac)
    FF_HD u192e prep(const u192e& cst) const { return montmul(cst, R2()); }
    FF_HD u192e mul(const u192e& a, const u192e& b) const { return montmul(montmul(a, b), R2()); }
    // a < 2^192: a * R^2 / R = a R mod p, then / R
    FF_HD u192e reduce_raw(const u192e& a) const {
        const u192e t = montmul(a, R2());
        const uint64_t x[6] = {t.lo, t.mid, t.hi, 0, 0, 0};
        return redc(x);
    }
    // T = y * x + cadd < 2^32 p + p as four limbs: REDC (T / R), then one Montgomery product by R^2
    FF_HD u192e muladd_small(const u192e& y, uint32_t x, const u192e& cadd) const {
        uint64_t t[6];
        ff_u128 v = (ff_u128)y.lo * x + cadd.lo;
        t[0] = ff_lo(v);
        v = (ff_u128)y.mid * x + cadd.mid + ff_hi(v);
        t[1] = ff_lo(v);
        v = (ff_u128)y.hi * x + cadd.hi + ff_hi(v);
        t[2] = ff_lo(v);
        t[3] = ff_hi(v);
        t[4] = t[5] = 0;
        return montmul(redc(t), R2());
    }
    FF_HD u192e muladd(const u192e& a, const u192e& b, const u192e& cadd) const { return add(mul(a, b), cadd); }


    FF_HD void acc_zero(acc& s) const { s.v.lo = s.v.mid = s.v.hi = 0; }
    FF_HD void acc_mac(acc& s, const u192e& lamR, const u192e& xe) const { s.v = add(s.v, montmul(lamR, xe)); }
    FF_HD u192e acc_reduce(const acc& s) const { return s.v; }
};


// ---------------------------------------------------------------------------
// GF2P8: GF(2^n), 1 <= n <= 8, one element per byte, arithmetic on four
// elements packed in a 32-bit word (SWAR shift-and-xor; gfpx.py:988-1003 _mul
// followed by :1025-1045 _mod, fused so intermediate degree never exceeds n).
// ---------------------------------------------------------------------------
struct GF2P8 {
    typedef uint8_t elem;
    typedef uint32_t word;
    enum { EPW = 4 };
    enum { BINARY = 1 };
    uint32_t n;     // extension degree
    uint32_t red;   // modulus without leading term, broadcast to 4 bytes
    uint32_t top;   // bit n-1 of every byte
    uint32_t emask; // low n bits of every byte

    struct acc {
        uint32_t a;
    };
    // constants are used as-is (no domain conversion)
    FF_HD uint32_t prep(uint32_t cst) const { return cst; }


    FF_HD uint32_t add(uint32_t a, uint32_t b) const { return a ^ b; }
    FF_HD uint32_t sub(uint32_t a, uint32_t b) const { return a ^ b; }
    FF_HD uint32_t neg(uint32_t a) const { return a; }
    FF_HD uint32_t reduce_raw(uint32_t a) const {
        // bytes may hold degree up to 7; reduce bits n..7 (only needed for n<8)
        if (n == 8) return a;
        uint32_t r = 0;
        // Horner over the 8 bits of every byte, MSB first
        for (int i = 7; i >= 0; --i) {
            r = xtime(r) ^ ((a >> i) & 0x01010101u);
        }
        return r;
    }
    FF_HD uint32_t xtime(uint32_t a) const {
        uint32_t hi = a & top;
        uint32_t m = hi >> (n - 1);          // 0/1 per byte
        uint32_t full = (hi - m) | hi;       // low n bits set where hi
        return ((a ^ hi) << 1) ^ (full & red);
    }
    FF_HD uint32_t mul(uint32_t a, uint32_t b) const {
        uint32_t c = 0;
        for (int i = (int)n - 1; i >= 0; --i) {
            uint32_t bm = (b >> i) & 0x01010101u;
            uint32_t t = bm << 7;
            uint32_t full = (t - bm) | t;    // 0xff where bit set
            c = xtime(c) ^ (a & full);
        }
        return c;
    }
    // y * X(x) + c where X(x) is the polynomial with bit pattern x (< 2^n)
    FF_HD uint32_t muladd_small(uint32_t y, uint32_t x, uint32_t cadd) const {
        // x < 2^n is wave-uniform (a party's x-coordinate, usually < 8): Horner over ITS bits only,
        // starting at its top set bit (scalar branches)
        if (x == 0) return cadd;
        uint32_t c = y;
        for (int i = 30 - __builtin_clz(x); i >= 0; --i) {
            c = xtime(c);
            if ((x >> i) & 1) c ^= y;
        }
        return c ^ cadd;
    }
    FF_HD uint32_t muladd(uint32_t a, uint32_t b, uint32_t cadd) const { return mul(a, b) ^ cadd; }
    FF_HD void acc_zero(acc& s) const { s.a = 0; }
    // lam: single element broadcast to the 4 bytes by the host
    FF_HD void acc_mac(acc& s, uint32_t lam, uint32_t x) const {
        // lam is wave-uniform: multiply x by the constant via scalar-branch Horner
        uint32_t l = lam & 0xffu;
        if (l == 0) return;
        uint32_t c = x;
        for (int i = 30 - __builtin_clz(l); i >= 0; --i) {
            c = xtime(c);
            if ((l >> i) & 1) c ^= x;
        }
        s.a ^= c;
    }
    FF_HD uint32_t acc_reduce(const acc& s) const { return s.a; }
};


// ---------------------------------------------------------------------------
// Carry-less multiplication through the integer multiplier.  gfx950 has no
// clmul instruction; a shift-xor loop costs ~25 VALU ops per BIT.  Instead the
// operands are split into four "every 4th bit" masks: in the integer product of
// two such masks every 4-bit slot receives at most 8 partial products, so
// nothing carries between slots and bit 0 of a slot is the GF(2) sum.  16
// v_mad_u64_u32 + ~46 logic ops give a 32x32 -> 64 carry-less product;
// Karatsuba (3 products per doubling) builds 64x64 and 128x128 from it.
// ---------------------------------------------------------------------------
// Three-input logic: gfx950's v_bitop3_b32 evaluates any 3-input truth table in one full-rate instruction.  The
// logic around the 16 multiplies of a 32x32 product is what bounds the wide binary fields (not the multiplies),
// so the device build spells out a ^ b ^ c and the bit select (m ? a : b): 16 + 6 logic ops per product instead
// of 24 + 14.  (Truth-table convention: the operands read as 0xF0, 0xCC, 0xAA.)  The host build (tests) uses the
// plain expressions.
#if defined(__HIP_DEVICE_COMPILE__)
FF_HD uint32_t ff_xor3(uint32_t a, uint32_t b, uint32_t c) { return __builtin_amdgcn_bitop3_b32(a, b, c, 0x96); }
FF_HD uint32_t ff_bsel(uint32_t m, uint32_t a, uint32_t b) { return __builtin_amdgcn_bitop3_b32(m, a, b, 0xCA); }
#else
FF_HD uint32_t ff_xor3(uint32_t a, uint32_t b, uint32_t c) { return a ^ b ^ c; }
FF_HD uint32_t ff_bsel(uint32_t m, uint32_t a, uint32_t b) { return (a & m) | (b & ~m); }
#endif
FF_HD uint64_t ff_xor3_64(uint64_t a, uint64_t b, uint64_t c) {
    return (uint64_t)ff_xor3((uint32_t)a, (uint32_t)b, (uint32_t)c) |
           ((uint64_t)ff_xor3((uint32_t)(a >> 32), (uint32_t)(b >> 32), (uint32_t)(c >> 32)) << 32);
}
FF_HD uint64_t ff_clmul32(uint32_t x, uint32_t y) {
    const uint32_t x0 = x & 0x11111111u, x1 = x & 0x22222222u, x2 = x & 0x44444444u, x3 = x & 0x88888888u;
    const uint32_t y0 = y & 0x11111111u, y1 = y & 0x22222222u, y2 = y & 0x44444444u, y3 = y & 0x88888888u;
    const uint64_t z0 = ff_xor3_64((uint64_t)x0 * y0, (uint64_t)x1 * y3, (uint64_t)x2 * y2) ^ ((uint64_t)x3 * y1);
    const uint64_t z1 = ff_xor3_64((uint64_t)x0 * y1, (uint64_t)x1 * y0, (uint64_t)x2 * y3) ^ ((uint64_t)x3 * y2);
    const uint64_t z2 = ff_xor3_64((uint64_t)x0 * y2, (uint64_t)x1 * y1, (uint64_t)x2 * y0) ^ ((uint64_t)x3 * y3);
    const uint64_t z3 = ff_xor3_64((uint64_t)x0 * y3, (uint64_t)x1 * y2, (uint64_t)x2 * y1) ^ ((uint64_t)x3 * y0);
    // slot bit k of every nibble comes from z_k
    const uint32_t lo = ff_bsel(0x11111111u, (uint32_t)z0, ff_bsel(0x22222222u, (uint32_t)z1, ff_bsel(0x44444444u, (uint32_t)z2, (uint32_t)z3)));
    const uint32_t hi = ff_bsel(0x11111111u, (uint32_t)(z0 >> 32),
                                ff_bsel(0x22222222u, (uint32_t)(z1 >> 32), ff_bsel(0x44444444u, (uint32_t)(z2 >> 32), (uint32_t)(z3 >> 32))));
    return (uint64_t)lo | ((uint64_t)hi << 32);
}
// 64 x 64 -> 128 (hi, lo)
FF_HD void ff_clmul64(uint64_t a, uint64_t b, uint64_t& hi, uint64_t& lo) {
    const uint32_t a0 = (uint32_t)a, a1 = (uint32_t)(a >> 32), b0 = (uint32_t)b, b1 = (uint32_t)(b >> 32);
    const uint64_t z0 = ff_clmul32(a0, b0);
    const uint64_t z2 = ff_clmul32(a1, b1);
    const uint64_t z1 = ff_xor3_64(ff_clmul32(a0 ^ a1, b0 ^ b1), z0, z2);
    lo = z0 ^ (z1 << 32);
    hi = z2 ^ (z1 >> 32);
}
// 128 x 128 -> 256 as four limbs p[0..3]
FF_HD void ff_clmul128(uint64_t alo, uint64_t ahi, uint64_t blo, uint64_t bhi, uint64_t p[4]) {
    uint64_t z0h, z0l, z2h, z2l, z1h, z1l;
    ff_clmul64(alo, blo, z0h, z0l);
    ff_clmul64(ahi, bhi, z2h, z2l);
    ff_clmul64(alo ^ ahi, blo ^ bhi, z1h, z1l);
    p[0] = z0l;
    p[1] = ff_xor3_64(z1l, z0l, z2l) ^ z0h;
    p[2] = ff_xor3_64(z1h, z0h, z2h) ^ z2l;
    p[3] = z2h;
}

// ---------------------------------------------------------------------------
// GF2W64 / GF2W128: GF(2^n) for 33 <= n <= 64 (and 9..32 for callers that ask for 8-byte words) / 65 <= n <= 128.
// mul: carry-less product (above) + reduction.  For moduli x^n + r(x) with a
// short r (r < 2^28: every default MPyC irreducible, e.g. x^128+x^7+x^2+x+1) the
// high part is folded down with one shift-xor per set bit of r (`fast`, nfold
// passes); other moduli take the bit-serial loop.  n <= 32 reduces bit-serially
// (at most 31 steps on a 64-bit product).
// ---------------------------------------------------------------------------
struct GF2W64 {
    typedef uint64_t elem;
    typedef uint64_t word;
    enum { EPW = 1 };
    enum { BINARY = 1 };
    uint64_t red;   // modulus without leading term
    uint64_t emask; // 2^n - 1
    uint32_t n;
    uint32_t fast;   // bit 0: sparse modulus (fold reduction); bits 8..15: number of fold passes
    struct acc {
        uint64_t a;
    };
    // constants are used as-is (no domain conversion)
    FF_HD uint64_t prep(uint64_t cst) const { return cst; }

    FF_HD uint64_t add(uint64_t a, uint64_t b) const { return a ^ b; }
    FF_HD uint64_t sub(uint64_t a, uint64_t b) const { return a ^ b; }
    FF_HD uint64_t neg(uint64_t a) const { return a; }
    FF_HD uint64_t xtime(uint64_t a) const {
        uint64_t hi = (a >> (n - 1)) & 1;
        return ((a << 1) & emask) ^ ((0 - hi) & red);
    }
    FF_HD uint64_t reduce_raw(uint64_t a) const {
        if (n == 64) return a;
        uint64_t r = 0;
        for (int i = 63; i >= 0; --i) r = xtime(r) ^ ((a >> i) & 1);
        return r;
    }
    FF_HD uint64_t mul_bitserial(uint64_t a, uint64_t b) const {
        uint64_t c = 0;
        for (int i = (int)n - 1; i >= 0; --i) {
            c = xtime(c) ^ ((0 - ((b >> i) & 1)) & a);
        }
        return c;
    }
    FF_HD uint64_t mul(uint64_t a, uint64_t b) const {
        if (n <= 32) {
            uint64_t pr = ff_clmul32((uint32_t)a, (uint32_t)b);
            // sparse modulus (every default MPyC irreducible): a few fold passes, as for the wider fields (round 6: GF(2^32)
            // took 31 long-division steps of ~10 instructions each after a 50-instruction product)
            if (fast & 1) return reduce_product32(pr);
            // otherwise long division by f = x^n + red from the top
            const uint64_t fm = red | (1ull << n);
            for (int i = 2 * (int)n - 2; i >= (int)n; --i) pr ^= (0 - ((pr >> i) & 1)) & (fm << (i - (int)n));
            return pr;
        }
        if (!(fast & 1)) return mul_bitserial(a, b);
        uint64_t hi, lo;
        ff_clmul64(a, b, hi, lo);
        return reduce_product(hi, lo);
    }
    // n <= 32: the product and every folded excess (h has fewer than n bits, red fewer than 28) fit one 64-bit word
    FF_HD uint64_t reduce_product32(uint64_t pr) const {
        const int folds = (int)((fast >> 8) & 0xff);
        for (int it = 0; it < folds; ++it) {
            const uint64_t h = pr >> n;
            pr &= emask;
            for (uint64_t rr = red; rr; rr &= rr - 1) pr ^= h << __builtin_ctzll(rr);
        }
        return pr;
    }
    // (hi:lo) = an unreduced carry-less product of two residues (degree <= 2n-2) -> residue; sparse moduli only (fast & 1)
    FF_HD uint64_t reduce_product(uint64_t hi, uint64_t lo) const {
        ff_u128 t = ff_make128(hi, lo);
        const int folds = (int)((fast >> 8) & 0xff);
        for (int it = 0; it < folds; ++it) {
            uint64_t h = (uint64_t)(t >> n);          // excess part (< 2^63)
            t &= (ff_u128)emask;
            for (uint64_t rr = red; rr; rr &= rr - 1) t ^= (ff_u128)h << __builtin_ctzll(rr);
        }
        return ff_lo(t);
    }
    FF_HD uint64_t muladd_small(uint64_t y, uint32_t x, uint64_t cadd) const {
        // x < min(2^32, 2^n) public and wave-uniform: Horner over its bits from the top set one
        if (x == 0) return cadd;
        uint64_t c = y;
        for (int i = 30 - __builtin_clz(x); i >= 0; --i) {
            c = xtime(c);
            if ((x >> i) & 1) c ^= y;
        }
        return c ^ cadd;
    }
    FF_HD uint64_t muladd(uint64_t a, uint64_t b, uint64_t cadd) const { return mul(a, b) ^ cadd; }
    FF_HD void acc_zero(acc& s) const { s.a = 0; }
    // (lam is wave-uniform; 1 -- the runtime's Lagrange coefficients at 0 for m in {3, 7} over GF(2^n) -- is a plain XOR)
    FF_HD void acc_mac(acc& s, uint64_t lam, uint64_t x) const { s.a ^= lam == 1ull ? x : mul(x, lam); }
    FF_HD uint64_t acc_reduce(const acc& s) const { return s.a; }
};

// GF2W32: GF(2^n) for 9 <= n <= 32 on FOUR-byte storage (round 6; rounds 1-5 kept these fields on the 8-byte words of GF2W64:
// correct, twice the bytes of every streaming kernel).  Same arithmetic: 32 x 32 carry-less product on the integer multiplier, fold
// passes for sparse moduli (`fast`), long division of the 64-bit product otherwise.
struct GF2W32 {
    typedef uint32_t elem;
    typedef uint32_t word;
    enum { EPW = 1 };
    enum { BINARY = 1 };
    uint32_t red;    // modulus without leading term
    uint32_t emask;  // 2^n - 1
    uint32_t n;      // 9..32
    uint32_t fast;   // bit 0: fold reduction; bits 8..15: number of fold passes
    struct acc {
        uint32_t a;
    };
    FF_HD uint32_t prep(uint32_t cst) const { return cst; }
    FF_HD uint32_t add(uint32_t a, uint32_t b) const { return a ^ b; }
    FF_HD uint32_t sub(uint32_t a, uint32_t b) const { return a ^ b; }
    FF_HD uint32_t neg(uint32_t a) const { return a; }
    FF_HD uint32_t xtime(uint32_t a) const {
        const uint32_t hi = (a >> (n - 1)) & 1u;
        return ((a << 1) & emask) ^ ((0u - hi) & red);
    }
    FF_HD uint32_t reduce_raw(uint32_t a) const {
        if (n == 32) return a;
        uint32_t r = 0;
        for (int i = 31; i >= 0; --i) r = xtime(r) ^ ((a >> i) & 1u);
        return r;
    }
    FF_HD uint32_t mul(uint32_t a, uint32_t b) const {
        uint64_t pr = ff_clmul32(a, b);
        if (fast & 1) {
            // the product and every folded excess (h has fewer than n bits, red fewer than 28) fit one 64-bit word
            const int folds = (int)((fast >> 8) & 0xff);
            for (int it = 0; it < folds; ++it) {
                const uint64_t h = pr >> n;
                pr &= (uint64_t)emask;
                for (uint32_t rr = red; rr; rr &= rr - 1) pr ^= h << __builtin_ctz(rr);
            }
            return (uint32_t)pr;
        }
        const uint64_t fm = (uint64_t)red | (1ull << n);          // long division by f = x^n + red from the top
        for (int i = 2 * (int)n - 2; i >= (int)n; --i) pr ^= (0 - ((pr >> i) & 1)) & (fm << (i - (int)n));
        return (uint32_t)pr;
    }
    FF_HD uint32_t muladd_small(uint32_t y, uint32_t x, uint32_t cadd) const {
        // x < 2^n public and wave-uniform: Horner over its bits from the top set one
        if (x == 0) return cadd;
        uint32_t c = y;
        for (int i = 30 - __builtin_clz(x); i >= 0; --i) {
            c = xtime(c);
            if ((x >> i) & 1) c ^= y;
        }
        return c ^ cadd;
    }
    FF_HD uint32_t muladd(uint32_t a, uint32_t b, uint32_t cadd) const { return mul(a, b) ^ cadd; }
    FF_HD void acc_zero(acc& s) const { s.a = 0; }
    // (lam is wave-uniform: the runtime's own Lagrange coefficients at 0 for m in {3, 7} parties are all 1 -- plain XOR of the rows)
    FF_HD void acc_mac(acc& s, uint32_t lam, uint32_t x) const { s.a ^= lam == 1u ? x : mul(x, lam); }
    FF_HD uint32_t acc_reduce(const acc& s) const { return s.a; }
};

struct GF2W128 {
    typedef u128e elem;
    typedef u128e word;
    enum { EPW = 1 };
    enum { BINARY = 1 };
    uint64_t red_lo, red_hi;      // modulus without leading term
    uint64_t emask_lo, emask_hi;  // 2^n - 1
    uint32_t n;                   // 65..128
    uint32_t fast;                // 1: modulus x^n + r, r < 2^28 (two-pass fold), else bit-serial
    struct acc {
        uint64_t lo, hi;
    };
    // constants are used as-is (no domain conversion)
    FF_HD u128e prep(u128e cst) const { return cst; }

    static FF_HD ff_u128 U(const u128e& a) { return ff_make128(a.hi, a.lo); }
    static FF_HD u128e E(ff_u128 x) {
        u128e r;
        r.lo = ff_lo(x);
        r.hi = ff_hi(x);
        return r;
    }
    FF_HD u128e add(const u128e& a, const u128e& b) const {
        u128e r;
        r.lo = a.lo ^ b.lo;
        r.hi = a.hi ^ b.hi;
        return r;
    }
    FF_HD u128e sub(const u128e& a, const u128e& b) const { return add(a, b); }
    FF_HD u128e neg(const u128e& a) const { return a; }
    FF_HD void xtime(uint64_t& lo, uint64_t& hi) const {
        uint64_t t = (hi >> (n - 65)) & 1;  // bit n-1
        uint64_t m = 0 - t;
        hi = ((hi << 1) | (lo >> 63)) & emask_hi;
        lo = lo << 1;
        lo ^= m & red_lo;
        hi ^= m & red_hi;
    }
    FF_HD u128e reduce_raw(const u128e& a) const {
        if (n == 128) return a;
        uint64_t lo = 0, hi = 0;
        for (int i = 127; i >= 0; --i) {
            xtime(lo, hi);
            uint64_t bit = i >= 64 ? (a.hi >> (i - 64)) & 1 : (a.lo >> i) & 1;
            lo ^= bit;
        }
        u128e r;
        r.lo = lo;
        r.hi = hi;
        return r;
    }
    FF_HD u128e mul(const u128e& a, const u128e& b) const {
        if (!(fast & 1)) return mul_bitserial(a, b);
        uint64_t p[4];
        ff_clmul128(a.lo, a.hi, b.lo, b.hi, p);
        return reduce_product(p);
    }
    // p[0..3] = an unreduced carry-less product of two residues (degree <= 2n-2) -> residue; sparse moduli only (fast & 1)
    FF_HD u128e reduce_product(const uint64_t p[4]) const {
        const uint32_t r = (uint32_t)red_lo;
        // H = P >> n (two limbs), L = P & mask;  T = L ^ H*r  (three limbs, H*r < 2^(n+27))
        uint64_t h0, h1;
        if (n == 128) {
            h0 = p[2];
            h1 = p[3];
        } else {
            const uint32_t sh = n - 64;       // 1..63
            h0 = (p[1] >> sh) | (p[2] << (64 - sh));
            h1 = (p[2] >> sh) | (p[3] << (64 - sh));
        }
        uint64_t t0 = p[0] & emask_lo, t1 = p[1] & emask_hi, t2 = 0;
        for (uint32_t rr = r; rr; rr &= rr - 1) {
            const int j = __builtin_ctz(rr);           // 0..27, wave-uniform
            t0 ^= h0 << j;
            t1 ^= (h1 << j) | (j ? (h0 >> (64 - j)) : 0);
            t2 ^= j ? (h1 >> (64 - j)) : 0;
        }
        // second pass: what stuck out above bit n is < 2^27
        uint64_t h2;
        if (n == 128) {
            h2 = t2;
        } else {
            const uint32_t sh = n - 64;
            h2 = (t1 >> sh) | (t2 << (64 - sh));
            t1 &= emask_hi;
        }
        for (uint32_t rr = r; rr; rr &= rr - 1) {
            const int j = __builtin_ctz(rr);
            t0 ^= h2 << j;                             // h2 * r < 2^54 < 2^n: stays in limb 0
        }
        u128e out;
        out.lo = t0;
        out.hi = t1;
        return out;
    }
    FF_HD u128e mul_bitserial(const u128e& a, const u128e& b) const {
        uint64_t lo = 0, hi = 0;
        for (int i = (int)n - 1; i >= 64; --i) {
            xtime(lo, hi);
            uint64_t m = 0 - ((b.hi >> (i - 64)) & 1);
            lo ^= m & a.lo;
            hi ^= m & a.hi;
        }
        for (int i = 63; i >= 0; --i) {
            xtime(lo, hi);
            uint64_t m = 0 - ((b.lo >> i) & 1);
            lo ^= m & a.lo;
            hi ^= m & a.hi;
        }
        u128e r;
        r.lo = lo;
        r.hi = hi;
        return r;
    }
    FF_HD u128e muladd_small(const u128e& y, uint32_t x, const u128e& cadd) const {
        if (x == 0) return cadd;
        uint64_t lo = y.lo, hi = y.hi;
        for (int i = 30 - __builtin_clz(x); i >= 0; --i) {
            xtime(lo, hi);
            if ((x >> i) & 1) {
                lo ^= y.lo;
                hi ^= y.hi;
            }
        }
        u128e r;
        r.lo = lo ^ cadd.lo;
        r.hi = hi ^ cadd.hi;
        return r;
    }
    FF_HD u128e muladd(const u128e& a, const u128e& b, const u128e& cadd) const {
        return add(mul(a, b), cadd);
    }
    FF_HD void acc_zero(acc& s) const { s.lo = s.hi = 0; }
    FF_HD void acc_mac(acc& s, const u128e& lam, const u128e& x) const {
        const u128e t = (lam.lo == 1ull && lam.hi == 0ull) ? x : mul(x, lam);      // (wave-uniform: 1 is a plain XOR)
        s.lo ^= t.lo;
        s.hi ^= t.hi;
    }
    FF_HD u128e acc_reduce(const acc& s) const {
        u128e r;
        r.lo = s.lo;
        r.hi = s.hi;
        return r;
    }
};

// ---- column accumulators for dot products with a SHARED operand (skinny products, matmul.hpp) ------------------
// Prime fields on one 64-bit word (PM64<*>, RC64).  acc_mac above forms the 128-bit product and adds it to a 192-bit sum:
// four v_mad_u64_u32 plus a chain of carries -- 32 instructions per term as the compiler writes it.  When the left
// operand a is the same for every lane (a row of activations against a matrix of weights) it is split ONCE into three
// limbs of 22 / 22 / 20 bits; with b = b1 2^32 + b0 the six partial products a_i b_j are below 2^54 and are summed per
// weight 2^(22 i + 32 j) in six 64-bit columns by the multiply-add itself: six v_mad_u64_u32 per term and nothing else,
// no carry for 2^10 terms.  gather() rebuilds the 192-bit sum for acc_reduce (its bound: at most 256 terms).
struct ColLimbs {
    uint32_t l0, l1, l2, pad;
};
FF_HD ColLimbs col_limbs(uint64_t a) {
    ColLimbs r;
    r.l0 = (uint32_t)a & 0x3fffffu;
    r.l1 = (uint32_t)(a >> 22) & 0x3fffffu;
    r.l2 = (uint32_t)(a >> 44);
    r.pad = 0;
    return r;
}
template <class ACC>
struct ColAcc {
    enum { MAX_TERMS = 256 };                                 // terms between two gather()s (the residue a flush puts back counts as one)
    uint64_t c00, c10, c20, c01, c11, c21;                    // c_ij: sum of a_i b_j, weight 2^(22 i + 32 j)
    FF_HD void zero() { c00 = c10 = c20 = c01 = c11 = c21 = 0; }
    FF_HD void mac(const ColLimbs& a, uint64_t b) {
        const uint32_t b0 = (uint32_t)b, b1 = (uint32_t)(b >> 32);
        c00 += (uint64_t)a.l0 * b0;
        c10 += (uint64_t)a.l1 * b0;
        c20 += (uint64_t)a.l2 * b0;
        c01 += (uint64_t)a.l0 * b1;
        c11 += (uint64_t)a.l1 * b1;
        c21 += (uint64_t)a.l2 * b1;
    }
    // <= 256 terms: c_ij < 2^62; the five columns below weight 2^64 sum to less than 2^117, c21 2^12 < 2^72 on top
    FF_HD ACC gather() const {
        const ff_u128 lo = (ff_u128)c00 + ((ff_u128)c10 << 22) + ((ff_u128)c01 << 32) + ((ff_u128)c20 << 44) + ((ff_u128)c11 << 54);
        const ff_u128 mid = (ff_u128)ff_hi(lo) + ((ff_u128)c21 << 12);
        ACC s;
        s.a0 = ff_lo(lo);
        s.a1 = ff_lo(mid);
        s.a2 = ff_hi(mid);
        return s;
    }
};
template <class F> struct col_mac_ok { enum { value = 0 }; };
template <bool K64, bool C1> struct col_mac_ok<PM64<K64, C1> > { enum { value = 1 }; };
template <> struct col_mac_ok<RC64> { enum { value = 1 }; };

// ---- share generation by forward differences (prime fields) -------------------------------------------------
// The parties' points are the consecutive integers 1..m (thresha.py:55-61), so f(x) = s + c_1 x + ... + c_T x^T is
// evaluated as f(x) = f(x-1) + D_1, D_1 += D_2, ..., D_{T-1} += D_T: T modular additions per share and no
// multiplication.  D_j = j! sum_i S(i, j) c_i at x = 0 with the Stirling numbers of the second kind (an identity
// in any commutative ring, so the residues are Horner's):
//   T = 1: D1 = c1             T = 2: D1 = c1 + c2, D2 = 2 c2          T = 3: D1 = c1 + c2 + c3, D2 = 2 c2 + 6 c3, D3 = 6 c3
//   T = 4: D1 = c1 + c2 + c3 + c4, D2 = 2 c2 + 6 c3 + 14 c4, D3 = 6 c3 + 36 c4, D4 = 24 c4
template <class F, int T>
FF_HD void share_diff_init(const F& f, const typename F::word (&c)[T], typename F::word (&dd)[T]) {
    typedef typename F::word W;
    static_assert(T >= 1 && T <= 4, "difference table written out for 1 <= T <= 4");
    if constexpr (T == 1) {
        dd[0] = c[0];
    } else if constexpr (T == 2) {
        dd[1] = f.add(c[1], c[1]);
        dd[0] = f.add(c[0], c[1]);
    } else if constexpr (T == 3) {
        const W c3x6 = f.muladd_small(c[2], 5u, c[2]);
        dd[2] = c3x6;
        dd[1] = f.add(f.add(c[1], c[1]), c3x6);
        dd[0] = f.add(f.add(c[0], c[1]), c[2]);
    } else {
        const W c3x6 = f.muladd_small(c[2], 5u, c[2]);
        dd[3] = f.muladd_small(c[3], 23u, c[3]);
        dd[2] = f.muladd_small(c[3], 36u, c3x6);
        dd[1] = f.muladd_small(c[3], 14u, f.add(f.add(c[1], c[1]), c3x6));
        dd[0] = f.add(f.add(c[0], c[1]), f.add(c[2], c[3]));
    }
}
// y = f(x - 1)  ->  f(x); the table moves on to x
template <class F, int T>
FF_HD typename F::word share_diff_next(const F& f, const typename F::word& y, typename F::word (&dd)[T]) {
    const typename F::word r = f.add(y, dd[0]);
#pragma unroll
    for (int j = 0; j + 1 < T; ++j) dd[j] = f.add(dd[j], dd[j + 1]);
    return r;
}

// ---- operand digits for the matrix-core dense product (kernels.hpp k_limb_gemm) ------------------------------
// L signed base-256 digits d_l in [-128, 127] of a representative of x modulo p.  L such digits represent exactly
// the integers in [-128 S, 127 S], S = (256^L - 1)/255 -- a window of 256^L - 1 >= p consecutive integers -- so the
// representative is x itself up to 127 S = 0x7f7f..7f and x - p above (not the balanced residue: for p close to
// 2^64 the value p/2 is NOT representable, the carries would run out of the top digit).
template <int L>
FF_HD uint64_t limb_digits_packed(uint64_t x, uint64_t p) {
    // All L digits at once: with C = 0x80...80, u = v + C adds 128 to every byte and lets the carries run, so byte l
    // of u is d_l + 128 and u ^ C holds the signed digits (sum_l (u_l - 128) 256^l = u - C = v; -C <= v <= top keeps
    // u inside 64 bits).  Digit l = byte l of the result; for L = 4 the upper bytes are the sign extension.
    const uint64_t top = 0x7f7f7f7f7f7f7f7full >> (8 * (8 - L));
    const uint64_t c = 0x8080808080808080ull >> (8 * (8 - L));
    const uint64_t v = (x > top) ? x - p : x;          // two's complement when negative
    return (v + c) ^ c;
}
template <int L>
FF_HD void limb_digits(uint64_t x, uint64_t p, int8_t (&d)[L]) {
    const uint64_t u = limb_digits_packed<L>(x, p);
#pragma unroll
    for (int l = 0; l < L; ++l) d[l] = (int8_t)(uint8_t)(u >> (8 * l));
}

// The same for two-limb values (primes of 65..128 bits): L = 12 (96-bit storage) or 16 digits.  x - p can be as
// low as -128 S ~ -0.502 * 2^128, so the running value is kept in three 64-bit words (two's complement).
template <int L>
FF_HD void limb_digits_wide(uint64_t xlo, uint64_t xhi, uint64_t plo, uint64_t phi, int8_t (&d)[L]) {
    static_assert(L > 8 && L <= 16, "two-limb operands have 9..16 digits");
    const uint64_t tlo = 0x7f7f7f7f7f7f7f7full, thi = 0x7f7f7f7f7f7f7f7full >> (8 * (16 - L));
    const bool above = xhi > thi || (xhi == thi && xlo > tlo);
    uint64_t w0 = xlo, w1 = xhi, w2 = 0;
    if (above) {                                   // v = x - p  (negative)
        const uint64_t b0 = xlo < plo;
        w0 = xlo - plo;
        const uint64_t t1 = xhi - phi, b1 = (xhi < phi) || (t1 < b0);
        w1 = t1 - b0;
        w2 = 0 - b1;
    }
#pragma unroll
    for (int l = 0; l < L; ++l) {
        const int8_t dl = (int8_t)(int)(w0 & 0xff);
        d[l] = dl;
        // v = (v >> 8) + (dl < 0), arithmetic shift over the three words
        w0 = (w0 >> 8) | (w1 << 56);
        w1 = (w1 >> 8) | (w2 << 56);
        w2 = (uint64_t)((int64_t)w2 >> 8);
        if (dl < 0) {
            if (++w0 == 0)
                if (++w1 == 0) ++w2;
        }
    }
}

