// rng.hpp -- on-device CSPRNG for share-generation coefficients.
//
// The reference draws each Shamir coefficient with secrets.randbelow(order) on the host
// (thresha.py:37,58-60), one OS-CSPRNG call per coefficient -- the dominant cost of its
// share generation.  Here coefficients come from a ChaCha keystream (block function of
// RFC 8439 / DJB ChaCha, 20 rounds by default) computed in registers by the thread that
// consumes them, keyed per call with 256 bits from the host's CSPRNG.
//
// Keystream -> field element ("sample"): see the Sampler<> specialisations below (rejection
// sampling for pseudo-Mersenne primes, W+64-bit wide samples for generic moduli, masks for GF(2^n)).
//
// Public layout (what tests/oracle reproduce): see RngLayout below -- groups of G adjacent 16-byte
// packs share B consecutive 64-byte blocks; samples are S bytes each in keystream order.
// Tail elements past the last full pack are drawn the same way with the pack index they
// would have had.
#pragma once
#include <stdint.h>
#include "fields.hpp"

namespace ffgpu {

struct RngKey {
    uint32_t key[8];    // 256-bit ChaCha key
    uint32_t nonce[2];  // stream id: unique per call (words 14, 15 of the state)
    uint32_t rounds;    // 20 (default), 12 or 8
    uint32_t pad_;
};

FF_HD uint32_t ff_rotl32(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }

#define FF_QR(a, b, c, d) \
    a += b; d ^= a; d = ff_rotl32(d, 16); \
    c += d; b ^= c; b = ff_rotl32(b, 12); \
    a += b; d ^= a; d = ff_rotl32(d, 8);  \
    c += d; b ^= c; b = ff_rotl32(b, 7);

// one 64-byte block; (w12,w13) = 64-bit block counter, (w14,w15) = nonce
FF_HD void chacha_block(const uint32_t key[8], uint32_t w12, uint32_t w13, uint32_t w14, uint32_t w15,
                        int rounds, uint32_t out[16]) {
    uint32_t x0 = 0x61707865u, x1 = 0x3320646eu, x2 = 0x79622d32u, x3 = 0x6b206574u;
    uint32_t x4 = key[0], x5 = key[1], x6 = key[2], x7 = key[3];
    uint32_t x8 = key[4], x9 = key[5], x10 = key[6], x11 = key[7];
    uint32_t x12 = w12, x13 = w13, x14 = w14, x15 = w15;
    for (int i = 0; i < rounds; i += 2) {
        FF_QR(x0, x4, x8, x12)
        FF_QR(x1, x5, x9, x13)
        FF_QR(x2, x6, x10, x14)
        FF_QR(x3, x7, x11, x15)
        FF_QR(x0, x5, x10, x15)
        FF_QR(x1, x6, x11, x12)
        FF_QR(x2, x7, x8, x13)
        FF_QR(x3, x4, x9, x14)
    }
    out[0] = x0 + 0x61707865u; out[1] = x1 + 0x3320646eu; out[2] = x2 + 0x79622d32u; out[3] = x3 + 0x6b206574u;
    out[4] = x4 + key[0]; out[5] = x5 + key[1]; out[6] = x6 + key[2]; out[7] = x7 + key[3];
    out[8] = x8 + key[4]; out[9] = x9 + key[5]; out[10] = x10 + key[6]; out[11] = x11 + key[7];
    out[12] = x12 + w12; out[13] = x13 + w13; out[14] = x14 + w14; out[15] = x15 + w15;
}

// Per-policy sampling.  S = bytes of keystream per sample.
//   * pseudo-Mersenne primes p = 2^k - c (PM64 / PM128 / PM96, every default MPyC prime): rejection
//     sampling, REJECT = 1.  A sample is k keystream bits; it is >= p with probability c/2^k (2^-56 for
//     2^64-189, 2^-61 for 2^61-1; at most 2^-17 for the smallest admissible prime, k = 33, c < 2^16).
//     A rejected sample is re-drawn from a keystream block OF ITS OWN -- block counter 2^63 + (global index
//     of the sample), same key and nonce; the main stream's counters stay below 2^63 -- whose 64/S candidate
//     samples are tried in order; only if all of them are >= p too (probability (c/2^k)^(1+64/S) < 2^-80 per
//     sample even for k = 33) is the last one conditionally reduced.  Two rejections therefore never receive
//     the same replacement (no shared spare samples), every accepted value is exactly uniform on [0, p), and
//     the only bias is that < 2^-80 event.  Half the keystream of the wide-sample method.
//   * generic moduli (RC64 / RC32 / MONT128): W + 64 uniform bits reduced mod p (bias < 2^-64, the
//     same idea as thresha.PRF's 16 extra bytes, thresha.py:234-236).  R = 2^W mod p comes from
//     the host (policy_build.hpp rng_const).
//   * GF(2^n): exact (mask to n bits).
// sample(f, R0, R1, w, &ok) -> canonical word (ok = 0: rejected, the caller re-draws with redraw()).
template <class F>
struct Sampler;

// candidates of the re-draw block of sample `sidx`
template <class Smp, class F>
FF_HD typename F::word rng_redraw(const F& f, const RngKey& rk, uint64_t sidx) {
    uint32_t blk[16];
    chacha_block(rk.key, (uint32_t)sidx, (uint32_t)(sidx >> 32) | 0x80000000u, rk.nonce[0], rk.nonce[1], (int)rk.rounds, blk);
    enum { NC = 64 / Smp::S };
    int ok;
    typename F::word r = Smp::last_resort(f, blk + (NC - 1) * (Smp::S / 4));
#pragma unroll
    for (int i = NC - 1; i >= 0; --i) {
        typename F::word v = Smp::sample(f, 0, 0, blk + i * (Smp::S / 4), &ok);
        if (ok) r = v;          // ends up with the FIRST candidate below p
    }
    return r;
}

template <bool K64, bool C1>
struct Sampler<PM64<K64, C1> > {
    enum { S = 8, REJECT = 1 };
    typedef PM64<K64, C1> F;
    static FF_HD uint64_t sample(const F& f, uint64_t, uint64_t, const uint32_t* w, int* ok) {
        uint64_t v = ((uint64_t)w[0] | ((uint64_t)w[1] << 32)) & f.mask;
        *ok = v < f.p;
        return v;
    }
    static FF_HD uint64_t last_resort(const F& f, const uint32_t* w) {
        return f.csub(((uint64_t)w[0] | ((uint64_t)w[1] << 32)) & f.mask);
    }
};

template <>
struct Sampler<RC64> {
    enum { S = 16, REJECT = 0 };
    static FF_HD uint64_t sample(const RC64& f, uint64_t R, uint64_t, const uint32_t* w, int*) {
        uint64_t lo = (uint64_t)w[0] | ((uint64_t)w[1] << 32);
        uint64_t hi = (uint64_t)w[2] | ((uint64_t)w[3] << 32);
        return f.add(f.mul(f.reduce_raw(hi), R), f.reduce_raw(lo));
    }
};

template <>
struct Sampler<RC32> {
    enum { S = 16, REJECT = 0 };  // 96 bits used, 32 skipped (keeps samples 16-byte aligned in the block)
    static FF_HD uint32_t sample(const RC32& f, uint64_t R, uint64_t, const uint32_t* w, int*) {
        uint32_t r = f.reduce_raw(w[2]);
        r = f.add(f.mul(r, (uint32_t)R), f.reduce_raw(w[1]));
        r = f.add(f.mul(r, (uint32_t)R), f.reduce_raw(w[0]));
        return r;
    }
};

template <bool K128>
struct Sampler<PM128<K128> > {
    enum { S = 16, REJECT = 1 };
    typedef PM128<K128> F;
    static FF_HD ff_u128 get(const F& f, const uint32_t* w) {
        uint64_t lo = (uint64_t)w[0] | ((uint64_t)w[1] << 32);
        uint64_t hi = (uint64_t)w[2] | ((uint64_t)w[3] << 32);
        return ff_make128(hi, lo) & f.M();
    }
    static FF_HD u128e sample(const F& f, uint64_t, uint64_t, const uint32_t* w, int* ok) {
        ff_u128 v = get(f, w);
        *ok = v < f.P();
        return F::E(v);
    }
    static FF_HD u128e last_resort(const F& f, const uint32_t* w) { return F::E(f.csub(get(f, w))); }
};

template <>
struct Sampler<PM96> {
    enum { S = 12, REJECT = 1 };   // three keystream words per sample
    typedef PM96 F;
    static FF_HD ff_u128 get(const F& f, const uint32_t* w) {
        uint64_t lo = (uint64_t)w[0] | ((uint64_t)w[1] << 32);
        return ff_make128((uint64_t)w[2], lo) & f.M();
    }
    static FF_HD u128e sample(const F& f, uint64_t, uint64_t, const uint32_t* w, int* ok) {
        ff_u128 v = get(f, w);
        *ok = v < f.P();
        return F::E(v);
    }
    static FF_HD u128e last_resort(const F& f, const uint32_t* w) { return F::E(f.csub(get(f, w))); }
};

template <>
struct Sampler<PM192> {
    enum { S = 24, REJECT = 1 };   // six keystream words per sample; two further candidates in a re-draw block
    typedef PM192 F;
    static FF_HD u192e get(const F& f, const uint32_t* w) {
        u192e v;
        v.lo = (uint64_t)w[0] | ((uint64_t)w[1] << 32);
        v.mid = (uint64_t)w[2] | ((uint64_t)w[3] << 32);
        v.hi = ((uint64_t)w[4] | ((uint64_t)w[5] << 32)) & f.mask_hi;
        return v;
    }
    static FF_HD u192e sample(const F& f, uint64_t, uint64_t, const uint32_t* w, int* ok) {
        const u192e v = get(f, w);
        *ok = !F::ge(v, f.P());
        return v;
    }
    static FF_HD u192e last_resort(const F& f, const uint32_t* w) { return f.csub(get(f, w)); }
};

template <>
struct Sampler<MONT128> {
    enum { S = 32, REJECT = 0 };
    static FF_HD u128e sample(const MONT128& f, uint64_t Rlo, uint64_t Rhi, const uint32_t* w, int*) {
        u128e lo, hi, R;
        lo.lo = (uint64_t)w[0] | ((uint64_t)w[1] << 32);
        lo.hi = (uint64_t)w[2] | ((uint64_t)w[3] << 32);
        hi.lo = (uint64_t)w[4] | ((uint64_t)w[5] << 32);
        hi.hi = (uint64_t)w[6] | ((uint64_t)w[7] << 32);
        R.lo = Rlo;
        R.hi = Rhi;
        return f.add(f.mul(f.reduce_raw(hi), R), f.reduce_raw(lo));
    }
};

template <>
struct Sampler<MONT192> {
    // 256 keystream bits per sample, reduced: (lo + 2^192 hi) mod p, lo of 192 bits, hi of 64 (bias < 2^-64 for
    // every 129..192-bit p).  2^192 hi mod p is ONE Montgomery product: montmul(hi, R^2) = hi R mod p, R = 2^192.
    enum { S = 32, REJECT = 0 };
    static FF_HD u192e sample(const MONT192& f, uint64_t, uint64_t, const uint32_t* w, int*) {
        u192e lo, hi;
        lo.lo = (uint64_t)w[0] | ((uint64_t)w[1] << 32);
        lo.mid = (uint64_t)w[2] | ((uint64_t)w[3] << 32);
        lo.hi = (uint64_t)w[4] | ((uint64_t)w[5] << 32);
        hi.lo = (uint64_t)w[6] | ((uint64_t)w[7] << 32);
        hi.mid = hi.hi = 0;
        return f.add(f.reduce_raw(lo), f.montmul(hi, f.R2()));
    }
};

template <>
struct Sampler<GF2P8> {
    enum { S = 4, REJECT = 0 };  // one 32-bit word = 4 packed elements, masked to n bits each
    static FF_HD uint32_t sample(const GF2P8& f, uint64_t, uint64_t, const uint32_t* w, int*) {
        return w[0] & f.emask;
    }
};
template <>
struct Sampler<GF2W32> {
    enum { S = 4, REJECT = 0 };
    static FF_HD uint32_t sample(const GF2W32& f, uint64_t, uint64_t, const uint32_t* w, int*) { return w[0] & f.emask; }
};
template <>
struct Sampler<GF2W64> {
    enum { S = 8, REJECT = 0 };
    static FF_HD uint64_t sample(const GF2W64& f, uint64_t, uint64_t, const uint32_t* w, int*) {
        return ((uint64_t)w[0] | ((uint64_t)w[1] << 32)) & f.emask;
    }
};
template <>
struct Sampler<GF2W128> {
    enum { S = 16, REJECT = 0 };
    static FF_HD u128e sample(const GF2W128& f, uint64_t, uint64_t, const uint32_t* w, int*) {
        u128e r;
        r.lo = ((uint64_t)w[0] | ((uint64_t)w[1] << 32)) & f.emask_lo;
        r.hi = ((uint64_t)w[2] | ((uint64_t)w[3] << 32)) & f.emask_hi;
        return r;
    }
};

// ---- keystream layout ---------------------------------------------------------------------------
// G packs (a "group") share B consecutive blocks: block counters group*B .. group*B+B-1.  With
// NG = ceil(npacks / G) groups, group g serves the packs g, g + NG, g + 2 NG, ... (a stride of NG, so
// that neighbouring lanes still touch neighbouring packs: every load stays fully coalesced);
// sample (u*NS + j*WPP + q) of the group belongs to pack g + u*NG, row j, word q; its GLOBAL index (the
// re-draw counter of a rejected sample) is g*G*NS + u*NS + j*WPP + q.  G in {1,2,3,4} is the value that
// wastes the least keystream (fewest blocks per pack, smallest G on ties) -- e.g. 64-bit primes: t=1 -> G=4
// (8 samples = one block for four packs), t=3 -> G=4 (24 samples = three blocks for four packs).
template <class F, int T, int WPP>
struct RngLayout {
    typedef Sampler<F> Smp;
    enum { S = Smp::S, NS = T * WPP };
    static constexpr int blocks(int g) { return (g * NS * S + 63) / 64; }
    static constexpr int best() {
        int bg = 1;
        for (int g = 2; g <= 4; ++g)
            if (blocks(g) * bg < blocks(bg) * g) bg = g;
        return bg;
    }
    enum { G = best(), B = blocks(best()) };
};

template <class F, int T, int WPP, bool R = (Sampler<F>::REJECT != 0)>
struct RedrawGroup {
    static FF_HD void run(const F&, const RngKey&, uint64_t, uint32_t, typename F::word[][T][WPP]) {}
};
template <class F, int T, int WPP>
struct RedrawGroup<F, T, WPP, true> {
    static FF_HD void run(const F& f, const RngKey& rk, uint64_t group, uint32_t rejected, typename F::word c[][T][WPP]) {
        typedef RngLayout<F, T, WPP> L;
#pragma unroll
        for (int idx = 0; idx < L::G * L::NS; ++idx)
            if ((rejected >> idx) & 1u)
                c[idx / L::NS][(idx % L::NS) / WPP][idx % WPP] =
                    rng_redraw<Sampler<F>, F>(f, rk, group * (uint64_t)(L::G * L::NS) + (uint64_t)idx);
    }
};
template <class F, int T, int WPP>
FF_HD void rng_redraw_group(const F& f, const RngKey& rk, uint64_t group, uint32_t rejected, typename F::word c[][T][WPP]) {
    RedrawGroup<F, T, WPP>::run(f, rk, group, rejected, c);
}

// draw all coefficients of group `group`: c[u][j][q]
template <class F, int T, int WPP>
FF_HD void rng_draw_group(const F& f, const RngKey& rk, uint64_t R0, uint64_t R1, uint64_t group,
                          typename F::word c[][T][WPP]) {
    typedef RngLayout<F, T, WPP> L;
    typedef Sampler<F> Smp;
    uint32_t ks[16 * L::B];
    const uint64_t ctr0 = group * (uint64_t)L::B;
#pragma unroll
    for (int b = 0; b < L::B; ++b) {
        const uint64_t ctr = ctr0 + (uint64_t)b;
        chacha_block(rk.key, (uint32_t)ctr, (uint32_t)(ctr >> 32), rk.nonce[0], rk.nonce[1], (int)rk.rounds,
                     ks + 16 * b);
    }
    static_assert(!Smp::REJECT || L::G * L::NS <= 32, "rejection mask is 32 bits");
    uint32_t rejected = 0;
#pragma unroll
    for (int u = 0; u < L::G; ++u)
#pragma unroll
        for (int sn = 0; sn < L::NS; ++sn) {
            int ok = 1;
            c[u][sn / WPP][sn % WPP] = Smp::sample(f, R0, R1, ks + (u * L::NS + sn) * (Smp::S / 4), &ok);
            if (Smp::REJECT && !ok) rejected |= 1u << ((u * L::NS + sn) & 31);
        }
    // re-draws happen after the whole group has been sampled: the keystream registers are dead by now, so the
    // (rare, divergent) extra block does not raise the kernel's register budget
    if (Smp::REJECT && rejected) rng_redraw_group<F, T, WPP>(f, rk, group, rejected, c);
}

// coefficients of ONE pack (scalar tails, unaligned inputs): draws the pack's group and selects
template <class F, int T, int WPP>
FF_HD void rng_draw_pack(const F& f, const RngKey& rk, uint64_t R0, uint64_t R1, uint64_t pack, uint64_t npacks,
                         typename F::word c[][WPP]) {
    typedef RngLayout<F, T, WPP> L;
    typename F::word g[L::G][T][WPP];
    const uint64_t ng = (npacks + L::G - 1) / L::G;
    rng_draw_group<F, T, WPP>(f, rk, R0, R1, pack % ng, g);
    const int u = (int)(pack / ng);
#pragma unroll
    for (int j = 0; j < T; ++j)
#pragma unroll
        for (int q = 0; q < WPP; ++q) {
            typename F::word v = g[0][j][q];
#pragma unroll
            for (int uu = 1; uu < L::G; ++uu) v = ff_pick(uu == u, g[uu][j][q], v);
            c[j][q] = v;
        }
}

}  // namespace ffgpu
