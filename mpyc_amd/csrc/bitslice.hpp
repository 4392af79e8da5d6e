// bitslice.hpp -- bit-sliced GF(2)[x] arithmetic on 32 elements at a time (round 4): 32 x 32 bit-matrix transposes,
// the product of N bit-planes by N bit-planes (Karatsuba down to 8 x 8 leaves, one partial-product MAC = one 3-input
// logic operation) and the fold modulo x^64 + x^4 + x^3 + x + 1.  Used by k_gf2w64_mul_bitsliced (misc.hip); compiles
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
// 32 elements (lo / hi words) -> the 32 products, through exactly the steps of the kernel
FF_HD void mul32(const uint32_t (&alo)[32], const uint32_t (&ahi)[32], const uint32_t (&blo)[32], const uint32_t (&bhi)[32],
                 uint32_t (&olo)[32], uint32_t (&ohi)[32]) {
    uint32_t pa[64], pb[64], c[127], t0[32], t1[32];
    for (int i = 0; i < 32; ++i) { t0[i] = alo[i]; t1[i] = ahi[i]; }
    transpose32(t0); transpose32(t1);
    for (int i = 0; i < 32; ++i) { pa[i] = t0[i]; pa[32 + i] = t1[i]; }
    for (int i = 0; i < 32; ++i) { t0[i] = blo[i]; t1[i] = bhi[i]; }
    transpose32(t0); transpose32(t1);
    for (int i = 0; i < 32; ++i) { pb[i] = t0[i]; pb[32 + i] = t1[i]; }
    Mul<64>::run(pa, pb, c);
    fold_1b(c);
    for (int i = 0; i < 32; ++i) { olo[i] = c[i]; ohi[i] = c[32 + i]; }
    transpose32(olo); transpose32(ohi);
}
}  // namespace bs64
}  // namespace ffgpu
