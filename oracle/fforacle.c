/*
 * oracle/fforacle.c -- plain-C restatement of the reference's hot path, for
 * full-size (10^7 element) parity checks and as the CPU baseline ("port").
 *
 * TEST INFRASTRUCTURE ONLY: only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg load liboracle; the product (mpyc_amd/, libffgpu.so) never does.
 *
 * Parity status: PINNED -- tests/test_oracle_golden.py checks these functions
 * against the tests/golden JSON fixtures generated from the real reference (see
 * tests/golden/make_golden.py), and against oracle/pyoracle.py.
 *
 * Deliberately textbook arithmetic (compiler-provided 128-bit `%` for one-limb
 * moduli; schoolbook 256-bit product + Knuth long division for two-limb moduli,
 * so that 128-bit fields run at OpenMP speed over 10^7 elements; shift-and-xor
 * for GF(2^n)); it shares no code and no reduction trick with the HIP kernels
 * (which fold by 2^k = c or use Montgomery products).
 *
 * Element layout: little-endian, width eb in {1,4,8,12,16} bytes, as include/ffgpu.h.
 * Reference lines restated (paths relative to the mpyc checkout):
 *   orc_ew        finfields.py:1056-1124,1189-1192 (+,-,*,neg), :717-725 (reduce),
 *                 gfpx.py:982-1045 (GF(2^n) add/mul/mod)
 *   orc_split     thresha.py:47-64 np_random_split
 *   orc_recombine thresha.py:119-132 np_recombine (vector from :67-85 supplied)
 *   orc_sbox      demos/np_aes.py:37-43 with runtime.py:1356-1367 (x^254 chain)
 */
#include <stddef.h>
#include <stdint.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef unsigned __int128 u128;

typedef struct {
    int binary;   /* 0: GF(p), 1: GF(2^n) */
    int eb;       /* element bytes */
    int n;        /* binary: degree */
    u128 p;       /* prime, or modulus without its leading term (binary) */
    u128 mask;    /* binary: 2^n - 1 */
} orc_field;

static int g_threads = 1;
void orc_set_threads(int t) { g_threads = t > 0 ? t : 1; }
int orc_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

static u128 mk(const uint64_t* l, int nl) { return ((u128)(nl > 1 ? l[1] : 0) << 64) | l[0]; }

int orc_field_init(orc_field* f, int binary, const uint64_t* mod, int nlimbs, int eb) {
    memset(f, 0, sizeof(*f));
    f->binary = binary;
    f->eb = eb;
    if (!binary) {
        f->p = mk(mod, nlimbs);
        return f->p >= 2 ? 0 : 1;
    }
    int deg = -1;
    if (nlimbs > 2 && mod[2]) deg = 128;
    else {
        u128 m = mk(mod, nlimbs);
        while (m) { ++deg; m >>= 1; }
    }
    if (deg < 1) return 1;
    f->n = deg;
    f->mask = deg == 128 ? ~(u128)0 : (((u128)1 << deg) - 1);
    f->p = mk(mod, nlimbs) & f->mask;   /* drop the leading term */
    return 0;
}

static inline u128 ld(const unsigned char* p, size_t i, int eb) {
    u128 v = 0;
    memcpy(&v, p + i * (size_t)eb, (size_t)eb);
    return v;
}
static inline void st(unsigned char* p, size_t i, int eb, u128 v) { memcpy(p + i * (size_t)eb, &v, (size_t)eb); }

static inline u128 addmod(u128 a, u128 b, u128 p) {
    u128 s = a + b;
    if (s < a || s >= p) s -= p;
    return s;
}
static inline u128 submod(u128 a, u128 b, u128 p) { return a >= b ? a - b : a - b + p; }

/* 128 x 128 -> 256-bit product, schoolbook on 64-bit halves: (hi, lo) */
static inline void mul256(u128 a, u128 b, u128* hi, u128* lo) {
    uint64_t a0 = (uint64_t)a, a1 = (uint64_t)(a >> 64), b0 = (uint64_t)b, b1 = (uint64_t)(b >> 64);
    u128 p00 = (u128)a0 * b0, p01 = (u128)a0 * b1, p10 = (u128)a1 * b0, p11 = (u128)a1 * b1;
    u128 mid = (p00 >> 64) + (uint64_t)p01 + (uint64_t)p10;              /* < 3 * 2^64 */
    *lo = (u128)(uint64_t)p00 | (mid << 64);
    *hi = p11 + (p01 >> 64) + (p10 >> 64) + (mid >> 64);
}

/* (hi:lo) mod p for a modulus of 65..128 bits: long division in base 2^64 (Knuth, TAOCP vol. 2, 4.3.1 algorithm D)
 * on a 4-digit dividend and a 2-digit divisor; only the remainder is kept.  Requires hi < p (true for a product of
 * two residues), which keeps every quotient digit below 2^64. */
static u128 mod256(u128 hi, u128 lo, u128 p) {
    uint64_t v1 = (uint64_t)(p >> 64), v0 = (uint64_t)p;
    int s = __builtin_clzll(v1);
    uint64_t u[5];
    if (s) {                                   /* D1: normalise so that the top divisor digit has its high bit set */
        v1 = (v1 << s) | (v0 >> (64 - s));
        v0 <<= s;
        u[4] = (uint64_t)(hi >> (128 - s));
        hi = (hi << s) | (lo >> (128 - s));
        lo <<= s;
    } else {
        u[4] = 0;
    }
    u[3] = (uint64_t)(hi >> 64); u[2] = (uint64_t)hi; u[1] = (uint64_t)(lo >> 64); u[0] = (uint64_t)lo;
    for (int j = 2; j >= 0; --j) {             /* D2..D7: one quotient digit per step */
        u128 num = ((u128)u[j + 2] << 64) | u[j + 1];
        u128 qhat, rhat;
        if (u[j + 2] >= v1) {                  /* quotient digit would overflow: start from base - 1 */
            qhat = ~(uint64_t)0;
            rhat = num - qhat * v1;
        } else {
            qhat = num / v1;
            rhat = num % v1;
        }
        while ((rhat >> 64) == 0 && qhat * v0 > ((rhat << 64) | u[j])) {      /* D3: at most two corrections */
            --qhat;
            rhat += v1;
        }
        /* D4: u[j..j+2] -= qhat * (v1:v0) */
        u128 prod0 = qhat * v0, prod1 = qhat * v1 + (prod0 >> 64);
        uint64_t s0 = (uint64_t)prod0, s1 = (uint64_t)prod1, s2 = (uint64_t)(prod1 >> 64);
        uint64_t b0 = u[j] < s0;
        u[j] -= s0;
        uint64_t t1 = u[j + 1] - s1, b1 = u[j + 1] < s1;
        uint64_t t1b = t1 - b0;
        b1 += t1 < b0;
        u[j + 1] = t1b;
        uint64_t t2 = u[j + 2] - s2, b2 = u[j + 2] < s2;
        uint64_t t2b = t2 - b1;
        b2 += t2 < b1;
        u[j + 2] = t2b;
        if (b2) {                              /* D6: qhat was one too large -> add the divisor back */
            u128 c = (u128)u[j] + v0;
            u[j] = (uint64_t)c;
            c = (u128)u[j + 1] + v1 + (c >> 64);
            u[j + 1] = (uint64_t)c;
            u[j + 2] += (uint64_t)(c >> 64);
        }
    }
    u128 r = ((u128)u[1] << 64) | u[0];        /* D8: denormalise the remainder */
    return s ? (r >> s) : r;
}

static inline u128 mulmod(u128 a, u128 b, u128 p) {
    if ((p >> 64) == 0) return (u128)(((u128)(uint64_t)a * (uint64_t)b) % p);
    u128 hi, lo;                               /* two-limb modulus: full product, then long division; a, b < p */
    mul256(a, b, &hi, &lo);
    return mod256(hi, lo, p);
}

/* the round-1/2 two-limb product (shift-and-add, one modular doubling per bit): kept as an independent cross-check of
 * mul256 + mod256 (tests/test_oracle_golden.py) */
u128 orc_mulmod_shift_add(u128 a, u128 b, u128 p) {
    u128 r = 0;
    while (b) {
        if (b & 1) r = addmod(r, a, p);
        a = addmod(a, a, p);
        b >>= 1;
    }
    return r;
}
void orc_mulmod_pair(const uint64_t* a, const uint64_t* b, const uint64_t* p, uint64_t* fast, uint64_t* slow) {
    u128 A = mk(a, 2), B = mk(b, 2), P = mk(p, 2);
    u128 f = mulmod(A, B, P), s = orc_mulmod_shift_add(A, B, P);
    fast[0] = (uint64_t)f; fast[1] = (uint64_t)(f >> 64);
    slow[0] = (uint64_t)s; slow[1] = (uint64_t)(s >> 64);
}

/* c = a*b in GF(2^n): MSB-first Horner, r stays below degree n */
static inline u128 gf2mul(const orc_field* f, u128 a, u128 b) {
    u128 r = 0;
    for (int i = f->n - 1; i >= 0; --i) {
        int carry = (int)((r >> (f->n - 1)) & 1);
        r = (r << 1) & f->mask;
        if (carry) r ^= f->p;
        if ((b >> i) & 1) r ^= a;
    }
    return r;
}
/* reduce an arbitrary bit pattern of width 8*eb */
static inline u128 gf2red(const orc_field* f, u128 a, int bits) {
    u128 r = 0;
    for (int i = bits - 1; i >= 0; --i) {
        int carry = (int)((r >> (f->n - 1)) & 1);
        r = (r << 1) & f->mask;
        if (carry) r ^= f->p;
        r ^= (a >> i) & 1;
    }
    return r;
}

static inline u128 f_add(const orc_field* f, u128 a, u128 b) { return f->binary ? a ^ b : addmod(a, b, f->p); }
static inline u128 f_sub(const orc_field* f, u128 a, u128 b) { return f->binary ? a ^ b : submod(a, b, f->p); }
static inline u128 f_mul(const orc_field* f, u128 a, u128 b) { return f->binary ? gf2mul(f, a, b) : mulmod(a, b, f->p); }
static inline u128 f_red(const orc_field* f, u128 a) { return f->binary ? gf2red(f, a, 8 * f->eb) : a % f->p; }

enum { ORC_ADD = 0, ORC_SUB = 1, ORC_MUL = 2, ORC_NEG = 3, ORC_REDUCE = 4 };

int orc_ew(const orc_field* f, int op, const unsigned char* a, const unsigned char* b, unsigned char* out,
           size_t n) {
    const int eb = f->eb;
    long long i;
#pragma omp parallel for num_threads(g_threads) schedule(static)
    for (i = 0; i < (long long)n; ++i) {
        u128 x = ld(a, (size_t)i, eb), y = b ? ld(b, (size_t)i, eb) : 0, r;
        switch (op) {
            case ORC_ADD: r = f_add(f, x, y); break;
            case ORC_SUB: r = f_sub(f, x, y); break;
            case ORC_MUL: r = f_mul(f, x, y); break;
            case ORC_NEG: r = f_sub(f, 0, x); break;
            default: r = f_red(f, x); break;
        }
        st(out, (size_t)i, eb, r);
    }
    return 0;
}

/* shares[i][h] = s[h] + sum_j C[j][h] * x_i^(j+1),  x_i = i+1 */
int orc_split(const orc_field* f, const unsigned char* s, const unsigned char* coef, size_t cstride, int t,
              int m, unsigned char* out, size_t ostride, size_t n) {
    const int eb = f->eb;
    long long h;
#pragma omp parallel for num_threads(g_threads) schedule(static)
    for (h = 0; h < (long long)n; ++h) {
        u128 sv = ld(s, (size_t)h, eb);
        for (int i = 0; i < m; ++i) {
            u128 x = f->binary ? (u128)(i + 1) : ((u128)(i + 1)) % f->p;
            u128 xp = 1, acc = sv;
            for (int j = 0; j < t; ++j) {
                xp = f_mul(f, xp, x);
                acc = f_add(f, acc, f_mul(f, ld(coef, (size_t)j * cstride + (size_t)h, eb), xp));
            }
            st(out, (size_t)i * ostride + (size_t)h, eb, acc);
        }
    }
    return 0;
}

/* out[r][h] = sum_j lam[r][j] * rows[j][h];  lam: (w,k) of two uint64 limbs */
int orc_recombine(const orc_field* f, const unsigned char* const* rows, const uint64_t* lam, int k, int w,
                  unsigned char* out, size_t ostride, size_t n) {
    const int eb = f->eb;
    long long h;
#pragma omp parallel for num_threads(g_threads) schedule(static)
    for (h = 0; h < (long long)n; ++h) {
        for (int r = 0; r < w; ++r) {
            u128 acc = 0;
            for (int j = 0; j < k; ++j) {
                const uint64_t* l = lam + 2 * ((size_t)r * k + j);
                acc = f_add(f, acc, f_mul(f, mk(l, 2), ld(rows[j], (size_t)h, eb)));
            }
            st(out, (size_t)r * ostride + (size_t)h, eb, acc);
        }
    }
    return 0;
}

/* C (M,N) = A (M,K) @ B (K,N): finfields.py:1126-1135 (object matmul, then one `%`) */
int orc_matmul(const orc_field* f, const unsigned char* A, const unsigned char* B, unsigned char* C, size_t M,
               size_t K, size_t N) {
    const int eb = f->eb;
    long long i;
#pragma omp parallel for num_threads(g_threads) schedule(static)
    for (i = 0; i < (long long)M; ++i)
        for (size_t j = 0; j < N; ++j) {
            u128 acc = 0;
            for (size_t k = 0; k < K; ++k)
                acc = f_add(f, acc, f_mul(f, ld(A, (size_t)i * K + k, eb), ld(B, k * N + j, eb)));
            st(C, (size_t)i * N + j, eb, acc);
        }
    return 0;
}

int orc_sbox(const unsigned char* in, const uint8_t* rows8, uint8_t b, unsigned char* out, size_t n) {
    orc_field f;
    uint64_t mod = 0x11b;
    orc_field_init(&f, 1, &mod, 1, 1);
    uint8_t lut[256];
    for (int v = 0; v < 256; ++v) {
        /* x^254 by the reference's chain (runtime.py:1356-1367) */
        u128 d = (u128)v, c = f_mul(&f, d, d), c2;
        c = f_mul(&f, c, c);
        c = f_mul(&f, c, c);
        c = f_mul(&f, c, d);
        c = f_mul(&f, c, c);
        c2 = f_mul(&f, c, c); d = f_mul(&f, c, d); c = c2;
        c2 = f_mul(&f, c, c); d = f_mul(&f, c, d); c = c2;
        c = f_mul(&f, c, d);
        c = f_mul(&f, c, c);
        unsigned iv = (unsigned)c, y = 0;
        for (int r = 0; r < 8; ++r) y |= (unsigned)(__builtin_popcount(iv & rows8[r]) & 1) << r;
        lut[v] = (uint8_t)(y ^ b);
    }
    long long i;
#pragma omp parallel for num_threads(g_threads) schedule(static)
    for (i = 0; i < (long long)n; ++i) out[i] = lut[in[i]];
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * Device CSPRNG restatement (mpyc_amd/csrc/rng.hpp).  The reference has no counterpart (it calls
 * secrets.randbelow per coefficient, thresha.py:37,58-60); this pins the *published algorithm*:
 * ChaCha block function (RFC 8439 section 2.3, checked against its test vector 2.3.2 in
 * tests/test_rng.py) + rejection sampling with per-sample re-draw blocks / "W+64 uniform bits mod p" sampling
 * + the documented keystream layout.
 * ------------------------------------------------------------------------------------------ */
static uint32_t rotl(uint32_t v, int c) { return (v << c) | (v >> (32 - c)); }
static void qround(uint32_t* s, int a, int b, int c, int d) {
    s[a] += s[b]; s[d] ^= s[a]; s[d] = rotl(s[d], 16);
    s[c] += s[d]; s[b] ^= s[c]; s[b] = rotl(s[b], 12);
    s[a] += s[b]; s[d] ^= s[a]; s[d] = rotl(s[d], 8);
    s[c] += s[d]; s[b] ^= s[c]; s[b] = rotl(s[b], 7);
}
void orc_chacha_block(const uint32_t key[8], const uint32_t w12_15[4], int rounds, uint32_t out[16]) {
    static const uint32_t sigma[4] = {0x61707865u, 0x3320646eu, 0x79622d32u, 0x6b206574u};
    uint32_t init[16], st[16];
    for (int i = 0; i < 4; ++i) init[i] = sigma[i];
    for (int i = 0; i < 8; ++i) init[4 + i] = key[i];
    for (int i = 0; i < 4; ++i) init[12 + i] = w12_15[i];
    memcpy(st, init, sizeof(st));
    for (int r = 0; r < rounds / 2; ++r) {
        qround(st, 0, 4, 8, 12); qround(st, 1, 5, 9, 13); qround(st, 2, 6, 10, 14); qround(st, 3, 7, 11, 15);
        qround(st, 0, 5, 10, 15); qround(st, 1, 6, 11, 12); qround(st, 2, 7, 8, 13); qround(st, 3, 4, 9, 14);
    }
    for (int i = 0; i < 16; ++i) out[i] = st[i] + init[i];
}

/* pseudo-Mersenne p = 2^k - c with c < 2^31 (and, below 2^64, the PM64 admissibility rule of
 * mpyc_amd/csrc/policy_build.hpp): these fields are sampled by rejection, all others wide. */
static int orc_is_pm(const orc_field* f, int* kbits) {
    if (f->binary) return 0;
    u128 p = f->p;
    int k = 0;
    while (k < 128 && (p >> k) > 0) ++k;   /* bit length */
    if (k < 33) return 0;
    u128 c = (k == 128) ? (u128)0 - p : (((u128)1) << k) - p;
    *kbits = k;
    if (k <= 64) {
        int cb = (k - 1) / 2 < 31 ? (k - 1) / 2 : 31;
        if (k == 64 && c == 1) return 0;
        return c < (((u128)1) << cb);
    }
    return c < (((u128)1) << 31);
}

static u128 ld_words(const uint32_t* w, int nwords) {
    u128 v = 0;
    for (int i = nwords - 1; i >= 0; --i) v = (v << 32) | w[i];
    return v;
}

/* coefficient matrix (t, n), row stride cstride elements, exactly as the device draws it */
int orc_rng_coeffs(const orc_field* f, const uint8_t key32[32], uint64_t nonce, int rounds, int t,
                   unsigned char* out, size_t cstride, size_t n) {
    const int eb = f->eb;
    const int EPV = 16 / eb;                         /* elements per 16-byte pack */
    const int packed = (f->binary && eb == 1);       /* GF(2^n<=8): a word holds 4 elements */
    const int WPP = packed ? 4 : EPV;                /* sampled words per pack */
    int kbits = 0;
    const int pm = orc_is_pm(f, &kbits);
    const int S = packed ? 4 : (f->binary ? eb : (pm ? eb : (eb == 16 ? 32 : 16)));
    const u128 kmask = kbits >= 128 ? ~(u128)0 : ((((u128)1) << kbits) - 1);
    uint32_t key[8];
    memcpy(key, key32, 32);
    if (rounds == 0) rounds = 20;
    const size_t npacks = (n + EPV - 1) / EPV;
    const int rows_per_draw = t <= 4 ? t : 1;
    const int draws = t <= 4 ? 1 : t;
    u128 R = 0;
    if (!f->binary && !pm && eb >= 8) {              /* 2^W mod p for the wide samples */
        R = 1;
        for (int i = 0; i < 8 * eb; ++i) R = addmod(R, R, f->p);
    }
    for (int d = 0; d < draws; ++d) {
        const int T = rows_per_draw;
        const int NS = T * WPP;
        /* group size: G in 1..4 with the fewest blocks per pack (smallest G on ties) */
        int G = 1;
        for (int g = 2; g <= 4; ++g) {
            int bg = (g * NS * S + 63) / 64, bG = (G * NS * S + 63) / 64;
            if (bg * G < bG * g) G = g;
        }
        const int B = (G * NS * S + 63) / 64;
        uint32_t n0 = (uint32_t)nonce, n1 = (uint32_t)(nonce >> 32);
        if (t > 4) n1 += (uint32_t)(d + 1);
        const size_t ngroups = (npacks + G - 1) / G;
        long long grp_;
#pragma omp parallel for num_threads(g_threads) schedule(static)      /* groups write disjoint elements */
        for (grp_ = 0; grp_ < (long long)ngroups; ++grp_) {
            const size_t grp = (size_t)grp_;
            uint32_t ks[16 * 32];
            for (int b = 0; b < B; ++b) {
                uint64_t ctr = (uint64_t)grp * B + b;
                uint32_t w[4] = {(uint32_t)ctr, (uint32_t)(ctr >> 32), n0, n1};
                orc_chacha_block(key, w, rounds, ks + 16 * b);
            }
            for (int u = 0; u < G; ++u) {
                size_t i = (size_t)u * ngroups + grp;   /* pack index: group g serves g, g+NG, g+2NG, ... */
                if (i >= npacks) continue;
                for (int sn = 0; sn < NS; ++sn) {
                    int j = sn / WPP, q = sn % WPP;
                    const uint32_t* w = ks + (u * NS + sn) * (S / 4);
                    int row = t > 4 ? d : j;
                    if (packed) {
                        uint32_t word = w[0] & (uint32_t)(0x01010101u * (uint32_t)f->mask);
                        for (int b_ = 0; b_ < 4; ++b_) {
                            size_t e = i * EPV + (size_t)q * 4 + b_;
                            if (e < n) st(out, (size_t)row * cstride + e, eb, (word >> (8 * b_)) & 0xff);
                        }
                        continue;
                    }
                    u128 v;
                    if (f->binary) {
                        v = ld_words(w, eb / 4) & f->mask;
                    } else if (pm) {
                        v = ld_words(w, eb / 4) & kmask;
                        if (v >= f->p) {
                            /* rejected: the sample's own re-draw block, counter 2^63 + its global index; first
                             * candidate below p wins, the last one is conditionally reduced if none is */
                            uint64_t sidx = (uint64_t)grp * (uint64_t)(G * NS) + (uint64_t)(u * NS + sn);
                            uint32_t wr[4] = {(uint32_t)sidx, (uint32_t)(sidx >> 32) | 0x80000000u, n0, n1};
                            uint32_t blk[16];
                            orc_chacha_block(key, wr, rounds, blk);
                            int nc = 64 / S, found = 0;
                            for (int ci = 0; ci < nc && !found; ++ci) {
                                v = ld_words(blk + ci * (S / 4), eb / 4) & kmask;
                                if (v < f->p) found = 1;
                            }
                            if (!found) v -= f->p;
                        }
                    } else if (eb == 4) {
                        v = ld_words(w, 3) % f->p;                      /* 96 bits */
                    } else if (eb == 8) {
                        u128 lo = ld_words(w, 2), hi = ld_words(w + 2, 2);
                        v = (mulmod(hi % f->p, R, f->p) + lo % f->p) % f->p;
                    } else {
                        u128 lo = ld_words(w, 4), hi = ld_words(w + 4, 4);
                        v = addmod(mulmod(hi % f->p, R, f->p), lo % f->p, f->p);
                    }
                    size_t e = i * EPV + (size_t)q;
                    if (e < n) st(out, (size_t)row * cstride + e, eb, v);
                }
            }
        }
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * PRSS in PRODUCTION mode (ffgpu_prss_chacha, mpyc_amd/csrc/kernels.hpp k_prss_chacha): the combination is the
 * reference's (thresha.py:163-173, 201-217: out[h] = sum_S sum_j W[S][j] * draw_S(h, j)), the PRF is a ChaCha stream
 * per subset key instead of SHAKE128 -- no reference counterpart, pinned to RFC 8439 + the layout below:
 *   LW = ceil(l / 4) words per draw; tile = TB consecutive blocks holding DPT = min(8, 16 TB / LW) draws, TB in
 *   {1, 2, 3} = the value with the most draws per block (smallest on ties); draw j of element h: tile h / DPT, slot
 *   h % DPT, block counters (tile * d + j) * TB + b; value = the l little-endian bytes at word slot * LW of the tile's
 *   keystream, `% p` (mask_bits == 0) or masked to mask_bits bits -- the reference's sampling rule (thresha.py:234-266).
 * keys40: ks x (32-byte key, 8-byte nonce); weights: ks * d scalars, two 64-bit limbs each.  Fields of up to 128 bits,
 * l <= 32 (wider ones: oracle/pyoracle.py).
 * ------------------------------------------------------------------------------------------ */
void orc_prss_chacha_layout(int l, int* tb, int* dpt) {
    const int lw = (l + 3) / 4;
    int best_tb = 0, best_dpt = 0;
    for (int t = 1; t <= 3; ++t) {
        int dp = 16 * t / lw;
        if (dp > 8) dp = 8;
        if (best_tb == 0 || dp * best_tb > best_dpt * t) { best_tb = t; best_dpt = dp; }
    }
    *tb = best_tb;
    *dpt = best_dpt;
}

int orc_prss_chacha(const orc_field* f, const uint8_t* keys40, int ks, int d, int l, int mask_bits, int rounds,
                    const uint64_t* weights, int accumulate, unsigned char* out, size_t n) {
    if (l < 1 || l > 32 || f->eb > 16) return 1;
    int tb, dpt;
    orc_prss_chacha_layout(l, &tb, &dpt);
    const int lw = (l + 3) / 4, eb = f->eb;
    const u128 r128 = f->binary ? 0 : ((~(u128)0) % f->p + 1) % f->p;      /* 2^128 mod p */
    long long h_;
#pragma omp parallel for num_threads(g_threads) schedule(static)
    for (h_ = 0; h_ < (long long)n; ++h_) {
        const size_t h = (size_t)h_;
        const uint64_t tile = h / (size_t)dpt;
        const int slot = (int)(h % (size_t)dpt);
        u128 acc = accumulate ? ld(out, h, eb) : 0;
        for (int s = 0; s < ks; ++s) {
            uint32_t key[8], nn[2];
            memcpy(key, keys40 + 40 * s, 32);
            memcpy(nn, keys40 + 40 * s + 32, 8);
            for (int j = 0; j < d; ++j) {
                uint32_t ksw[16 * 3];
                for (int b = 0; b < tb; ++b) {
                    uint64_t ctr = (tile * (uint64_t)d + (uint64_t)j) * (uint64_t)tb + (uint64_t)b;
                    uint32_t w[4] = {(uint32_t)ctr, (uint32_t)(ctr >> 32), nn[0], nn[1]};
                    orc_chacha_block(key, w, rounds, ksw + 16 * b);
                }
                unsigned char bytes[32];
                memset(bytes, 0, sizeof(bytes));
                memcpy(bytes, ksw + slot * lw, (size_t)l);        /* little-endian host: words -> bytes in keystream order */
                u128 lo, hi;
                memcpy(&lo, bytes, 16);
                memcpy(&hi, bytes + 16, 16);
                u128 v;
                if (mask_bits > 0) {
                    v = mask_bits >= 128 ? lo : (lo & ((((u128)1) << mask_bits) - 1));
                } else {                                           /* (hi 2^128 + lo) mod p, any p of up to 128 bits */
                    v = addmod(mulmod(hi % f->p, r128, f->p), lo % f->p, f->p);
                }
                u128 wt = ((u128)weights[2 * (s * d + j) + 1] << 64) | weights[2 * (s * d + j)];
                acc = f_add(f, acc, f_mul(f, wt, f_red(f, v)));
            }
        }
        st(out, h, eb, acc);
    }
    return 0;
}

size_t orc_field_sizeof(void) { return sizeof(orc_field); }
