// bitslice.hpp -- bit-sliced GF(2)[x] arithmetic: 32 x 32 bit-matrix transposes, the product of N bit-planes by N bit-planes
// (Karatsuba down to 8 x 8 leaves, one partial-product MAC = one 3-input logic operation), the fold modulo
// x^64 + x^4 + x^3 + x + 1, and the packed 64 x 64 product on 16 elements per lane (mul16_planes).  Used by k_gf2w64_mul_bitsliced (misc.hip); compiles
// with g++ as well (FF_HD), which is how tests/test_hostcheck.py checks it against a bit-serial product without a GPU.
// Mirrors gfpx.py:988-1045 (BinaryPolynomial._mul / _mod) for the default GF(2^64) modulus.
#pragma once
#include "fields.hpp"

namespace ffgpu {
namespace bs64 {
#if defined(__HIP_DEVICE_COMPILE__)
FF_HD uint32_t mac(uint32_t acc, uint32_t a, uint32_t b) { return __builtin_amdgcn_bitop3_b32(acc, a, b, 0x78); }   // acc ^ (a & b)
FF_HD uint32_t perm(uint32_t hi, uint32_t lo, uint32_t sel) { return __builtin_amdgcn_perm(hi, lo, sel); }
#else
FF_HD uint32_t mac(uint32_t acc, uint32_t a, uint32_t b) { return acc ^ (a & b); }
FF_HD uint32_t perm(uint32_t hi, uint32_t lo, uint32_t sel) {        // v_perm_b32: byte k of the result = byte sel_k of {hi, lo}
    const uint64_t src = ((uint64_t)hi << 32) | lo;
    uint32_t r = 0;
    for (int k = 0; k < 4; ++k) r |= (uint32_t)((src >> (8 * ((sel >> (8 * k)) & 7))) & 0xffu) << (8 * k);
    return r;
}
#endif
template <int S>
FF_HD void tr_stage(uint32_t (&A)[32]) {
    constexpr uint32_t M = S == 4 ? 0x0f0f0f0fu : S == 2 ? 0x33333333u : 0x55555555u;
#pragma unroll
    for (int k = 0; k < 32; ++k) {
        if (k & S) continue;
        const uint32_t x = A[k], y = A[k + S];
        if constexpr (S == 16) {
            A[k] = perm(y, x, 0x05040100u);          // lo16(x) | lo16(y) << 16
            A[k + S] = perm(y, x, 0x07060302u);      // hi16(x) | hi16(y) << 16
        } else if constexpr (S == 8) {
            A[k] = perm(y, x, 0x06020400u);          // bytes x0, y0, x2, y2
            A[k + S] = perm(y, x, 0x07030501u);      // bytes x1, y1, x3, y3
        } else {
            A[k] = ff_bsel(M, x, y << S);
            A[k + S] = ff_bsel(M, x >> S, y);
        }
    }
}
// 32 x 32 bit-matrix transpose in registers: out word i, bit e = in word e, bit i
FF_HD void transpose32(uint32_t (&A)[32]) {
    tr_stage<16>(A); tr_stage<8>(A); tr_stage<4>(A); tr_stage<2>(A); tr_stage<1>(A);
}
// c (2N - 1 planes) = a (N planes) x b (N planes) over GF(2)[x]
template <int N>
struct Mul {
    static FF_HD void run(const uint32_t* a, const uint32_t* b, uint32_t* c) {
        constexpr int H = N / 2;
        uint32_t z0[N - 1], z2[N - 1], zm[N - 1], am[H], bm[H];
        Mul<H>::run(a, b, z0);
        Mul<H>::run(a + H, b + H, z2);
#pragma unroll
        for (int i = 0; i < H; ++i) { am[i] = a[i] ^ a[i + H]; bm[i] = b[i] ^ b[i + H]; }
        Mul<H>::run(am, bm, zm);
#pragma unroll
        for (int k = 0; k < 2 * N - 1; ++k) {
            uint32_t v = k < N - 1 ? z0[k] : (k >= N ? z2[k - N] : 0u);
            const int q = k - H;
            if (q >= 0 && q < N - 1) {
                const uint32_t mid = ff_xor3(zm[q], z0[q], z2[q]);
                v = (k == N - 1) ? mid : (v ^ mid);
            }
            c[k] = v;
        }
    }
};
template <>
struct Mul<8> {
    static FF_HD void run(const uint32_t* a, const uint32_t* b, uint32_t* c) {
#pragma unroll
        for (int k = 0; k < 15; ++k) {
            uint32_t acc = 0;
            bool first = true;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int j = k - i;
                if (j < 0 || j > 7) continue;
                acc = first ? (a[i] & b[j]) : mac(acc, a[i], b[j]);
                first = false;
            }
            c[k] = acc;
        }
    }
};
// 127 product planes -> 64 planes modulo x^64 + x^4 + x^3 + x + 1 (0x1b): four XORs per high plane, highest first
FF_HD void fold_1b(uint32_t (&c)[127]) {
#pragma unroll
    for (int k = 126; k >= 64; --k) {
        const uint32_t h = c[k];
        c[k - 64] ^= h; c[k - 63] ^= h; c[k - 61] ^= h; c[k - 60] ^= h;
    }
}

// ---- the 64 x 64 product on 16 elements per lane, the two HALVES of an operand packed into one register (round 5) ------
// An element a = A0 + x^32 A1.  Plane register i (i < 32) holds coefficient i of A0 of the lane's 16 elements in its low
// half and coefficient i of A1 in its high half -- which is exactly what ONE 32 x 32 transpose of the rows [lo words of the
// 16 elements ; hi words] produces.  Every logic instruction then works on two polynomials at once:
//   P = Mul<32>(pa, pb)    low halves: z0 = A0 B0, high halves: z2 = A1 B1      (the two outer Karatsuba products)
// and the middle product zm = (A0 + A1)(B0 + B1), a 32 x 32 product of 16-bit planes, is packed the same way one level
// down -- (M0 | M1), M = A0 + A1 = M0 + x^16 M1 -- and again at 16 x 16; only the last 8 x 8 leaf runs half empty.
// 9 + 3 + 1 + 1 leaves for 16 elements against 27 for 32: 4 % more MACs, and HALF the registers: no operand or product of the
// 64 x 64 level is ever held as 64 or 127 full-width planes (the kernel fits two waves per SIMD, which is what the
// one-wave kernel of round 4 lacked: its VALU was busy 0.65 of the time).  `>> 16` moves a high half under a low half;
// garbage above bit 15 is never read.
FF_HD uint32_t lo_pair(uint32_t hi_src, uint32_t lo_src) { return perm(hi_src, lo_src, 0x05040100u); }   // lo16(lo_src) | lo16(hi_src) << 16
template <int N>
FF_HD void fold_halves(const uint32_t (&x)[2 * N], uint32_t (&y)[N]) {          // y_i = lo16(x_i ^ hi(x_i)) | lo16(x_{i+N} ^ hi(x_{i+N})) << 16
#pragma unroll
    for (int i = 0; i < N; ++i) y[i] = lo_pair(x[i + N] ^ (x[i + N] >> 16), x[i] ^ (x[i] >> 16));
}
// inner (2N-1 planes, low halves) of the Karatsuba step whose outer products are packed in `outer` (2N-1 registers: low = z0,
// high = z2) and whose middle product is `mid` (2N-1, low halves): result 4N-1 planes, low halves
template <int N>
FF_HD void combine_packed(const uint32_t (&outer)[2 * N - 1], const uint32_t (&mid)[2 * N - 1], uint32_t (&c)[4 * N - 1]) {
#pragma unroll
    for (int k = 0; k < 4 * N - 1; ++k) {
        uint32_t v = 0;
        if (k < 2 * N - 1) v = outer[k];
        if (k >= 2 * N) v ^= outer[k - 2 * N] >> 16;
        const int q = k - N;
        if (q >= 0 && q < 2 * N - 1) v ^= ff_xor3(mid[q], outer[q], outer[q] >> 16);
        c[k] = v;
    }
}
// pa, pb: 32 packed plane registers each (see above) -> c: the 127 planes of the product, low halves
FF_HD void mul16_planes(const uint32_t (&pa)[32], const uint32_t (&pb)[32], uint32_t (&c)[127]) {
    uint32_t zm[63];
    {
        uint32_t qa[16], qb[16], ym[31];
        fold_halves<16>(pa, qa);
        fold_halves<16>(pb, qb);
        {
            uint32_t ra[8], rb[8], R[15], S[15];
            fold_halves<8>(qa, ra);
            fold_halves<8>(qb, rb);
            Mul<8>::run(ra, rb, R);                          // low: w0, high: w2
            uint32_t sa[8], sb[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) { sa[i] = ra[i] ^ (ra[i] >> 16); sb[i] = rb[i] ^ (rb[i] >> 16); }
            Mul<8>::run(sa, sb, S);                          // low: wm (the one half-empty leaf)
            combine_packed<8>(R, S, ym);
        }
        uint32_t Q[31];
        Mul<16>::run(qa, qb, Q);                             // low: y0 = M0 N0, high: y2 = M1 N1
        combine_packed<16>(Q, ym, zm);
    }
    uint32_t P[63];
    Mul<32>::run(pa, pb, P);                                 // low: z0 = A0 B0, high: z2 = A1 B1
    combine_packed<32>(P, zm, c);
}
// 16 elements (lo / hi words) -> the 16 products modulo x^64 + x^4 + x^3 + x + 1, through exactly the steps of the kernel
FF_HD void mul16_packed(const uint32_t (&alo)[16], const uint32_t (&ahi)[16], const uint32_t (&blo)[16], const uint32_t (&bhi)[16],
                        uint32_t (&olo)[16], uint32_t (&ohi)[16]) {
    uint32_t pa[32], pb[32], c[127], o[32];
    for (int e = 0; e < 16; ++e) { pa[e] = alo[e]; pa[16 + e] = ahi[e]; pb[e] = blo[e]; pb[16 + e] = bhi[e]; }
    transpose32(pa); transpose32(pb);
    mul16_planes(pa, pb, c);
    fold_1b(c);
    for (int i = 0; i < 32; ++i) o[i] = lo_pair(c[i + 32], c[i]);
    transpose32(o);
    for (int e = 0; e < 16; ++e) { olo[e] = o[e]; ohi[e] = o[16 + e]; }
}
}  // namespace bs64
}  // namespace ffgpu
