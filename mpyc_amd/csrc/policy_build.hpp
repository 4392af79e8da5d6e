// policy_build.hpp -- host-only: classify a modulus and build the field policy
// (fields.hpp) that the kernels receive by value.  Shared by api.hip and by the
// g++-compiled arithmetic check in tests/hostcheck.cpp, so both see identical
// constants.
#pragma once
#include <stdint.h>
#include <string.h>
#include "fields.hpp"

namespace ffgpu {

enum PolicyKind {
    POL_NONE = 0,
    POL_PM64_MERSENNE,  // PM64<false,true>
    POL_PM64_K64,       // PM64<true,false>
    POL_PM64_GEN,       // PM64<false,false>
    POL_RC64,
    POL_RC32,
    POL_PM128_K128,     // PM128<true>
    POL_PM128_GEN,      // PM128<false>
    POL_PM96,           // PM96 (12-byte storage)
    POL_MONT128,
    POL_GF2P8,
    POL_GF2W64,
    POL_GF2W128,
    POL_PM192,          // three-limb pseudo-Mersenne primes (24-byte storage)
    POL_MONT192,        // any other odd prime of 129..192 bits (24-byte storage)
    POL_GF2W32          // GF(2^n), 9 <= n <= 32, 4-byte storage (round 6)
};

// status values mirror include/ffgpu.h
enum { PB_OK = 0, PB_EINVAL = 1, PB_ENOTSUP = 2, PB_EMODULUS = 4 };
// reduction tags mirror FFGPU_RED_*
enum { PB_RED_PM = 1, PB_RED_RC = 2, PB_RED_SWAR = 3, PB_RED_WIDE = 4, PB_RED_MONT = 5 };

struct PolicyBlob {
    int kind;
    int reduction;
    int elem_bytes;
    alignas(16) unsigned char bytes[128];
};

template <class F>
inline void store_policy(PolicyBlob* c, const F& f, int kind, int red) {
    static_assert(sizeof(F) <= sizeof(c->bytes), "policy blob too small");
    memset(c->bytes, 0, sizeof(c->bytes));
    memcpy(c->bytes, &f, sizeof(F));
    c->kind = kind;
    c->reduction = red;
    c->elem_bytes = (int)sizeof(typename F::elem);
}

inline int bitlen128(ff_u128 x) {
    int n = 0;
    while (x) {
        ++n;
        x >>= 1;
    }
    return n;
}

inline int build_prime_policy(PolicyBlob* c, ff_u128 p) {
    if (p < 2) return PB_EMODULUS;
    int k = bitlen128(p);
    if (k <= 32) {
        RC32 f;
        f.p = (uint32_t)p;
        f.s = (uint32_t)__builtin_clz(f.p);
        f.d = f.p << f.s;
        f.v = (uint32_t)(0xFFFFFFFFFFFFFFFFull / f.d);
        store_policy(c, f, POL_RC32, PB_RED_RC);
        return PB_OK;
    }
    if (k <= 64) {
        uint64_t p64 = (uint64_t)p;
        ff_u128 cc = ((ff_u128)1 << k) - p;
        int cb = (k - 1) / 2 < 31 ? (k - 1) / 2 : 31;
        bool k64 = (k == 64);
        if (cc < ((ff_u128)1 << cb) && !(k64 && cc == 1)) {
            uint64_t mask = k64 ? ~0ull : ((1ull << k) - 1);
            if (!k64 && cc == 1 && k <= 61) {             // (k = 62, 63 would overflow PM64<.,true>::presum: general path)
                PM64<false, true> f;
                f.p = p64; f.mask = mask; f.c = 1; f.k = (uint32_t)k;
                store_policy(c, f, POL_PM64_MERSENNE, PB_RED_PM);
            } else if (k64) {
                PM64<true, false> f;
                f.p = p64; f.mask = mask; f.c = (uint32_t)cc; f.k = 64;
                store_policy(c, f, POL_PM64_K64, PB_RED_PM);
            } else {
                PM64<false, false> f;
                f.p = p64; f.mask = mask; f.c = (uint32_t)cc; f.k = (uint32_t)k;
                store_policy(c, f, POL_PM64_GEN, PB_RED_PM);
            }
            return PB_OK;
        }
        RC64 f;
        f.p = p64;
        f.s = (uint32_t)__builtin_clzll(p64);
        f.d = p64 << f.s;
        f.v = (uint64_t)((~(ff_u128)0) / f.d);
        f.pad_ = 0;
        store_policy(c, f, POL_RC64, PB_RED_RC);
        return PB_OK;
    }
    // two limbs
    ff_u128 cc = (k == 128) ? (ff_u128)0 - p : ((ff_u128)1 << k) - p;
    if (cc < ((ff_u128)1 << 31)) {
        ff_u128 mask = (k == 128) ? ~(ff_u128)0 : (((ff_u128)1 << k) - 1);
        if (k == 128) {
            PM128<true> f;
            f.p_lo = ff_lo(p); f.p_hi = ff_hi(p); f.mask_lo = ff_lo(mask); f.mask_hi = ff_hi(mask);
            f.c = (uint32_t)cc; f.k = 128;
            store_policy(c, f, POL_PM128_K128, PB_RED_PM);
        } else if (k <= 96) {
            PM96 f;
            f.p_lo = ff_lo(p); f.p_hi = ff_hi(p); f.mask_lo = ff_lo(mask); f.mask_hi = ff_hi(mask);
            f.c = (uint32_t)cc; f.k = (uint32_t)k;
            store_policy(c, f, POL_PM96, PB_RED_PM);
        } else {
            PM128<false> f;
            f.p_lo = ff_lo(p); f.p_hi = ff_hi(p); f.mask_lo = ff_lo(mask); f.mask_hi = ff_hi(mask);
            f.c = (uint32_t)cc; f.k = (uint32_t)k;
            store_policy(c, f, POL_PM128_GEN, PB_RED_PM);
        }
        return PB_OK;
    }
    if (!(p & 1)) return PB_EMODULUS;
    MONT128 f;
    f.p_lo = ff_lo(p);
    f.p_hi = ff_hi(p);
    // -p^{-1} mod 2^64 by Newton iteration
    uint64_t inv = f.p_lo;  // correct to 3 bits
    for (int i = 0; i < 6; ++i) inv *= 2 - f.p_lo * inv;
    f.pinv = 0 - inv;
    f.pad_ = 0;
    // R^2 = 2^256 mod p by 256 modular doublings of 1
    ff_u128 r = 1;
    for (int i = 0; i < 256; ++i) r = f.addu(r, r);
    f.r2_lo = ff_lo(r);
    f.r2_hi = ff_hi(r);
    store_policy(c, f, POL_MONT128, PB_RED_MONT);
    return PB_OK;
}

// primes given as up to three limbs: 129..192-bit primes of the shape 2^k - c, c < 2^31, get the PM192 policy, all
// other odd ones the three-limb Montgomery policy
inline int build_prime_policy3(PolicyBlob* c, const uint64_t* mod, int nlimbs) {
    const uint64_t m2 = nlimbs > 2 ? mod[2] : 0;
    if (!m2) return build_prime_policy(c, ff_make128(nlimbs > 1 ? mod[1] : 0, mod[0]));
    const int k = 128 + (64 - __builtin_clzll(m2));
    // 2^k - p must be < 2^31: all bits of p above bit 31 and below bit k are ones
    const uint64_t top_mask = k == 192 ? ~0ull : ((1ull << (k - 128)) - 1);
    const uint64_t cc = (0 - mod[0]) & 0xffffffffull;            // 2^k - p = 2^64 - mod[0] (the upper limbs are all ones)
    if (m2 != top_mask || mod[1] != ~0ull || (mod[0] >> 31) != (~0ull >> 31) || cc == 0 || cc >= (1ull << 31)) {
        if (!(mod[0] & 1)) return PB_EMODULUS;
        MONT192 g;
        g.p0 = mod[0];
        g.p1 = mod[1];
        g.p2 = m2;
        uint64_t inv = g.p0;                                     // -p^{-1} mod 2^64 by Newton iteration
        for (int i = 0; i < 6; ++i) inv *= 2 - g.p0 * inv;
        g.pinv = 0 - inv;
        g.pad_ = 0;
        u192e r;                                                 // R^2 = 2^384 mod p by 384 modular doublings of 1
        r.lo = 1;
        r.mid = r.hi = 0;
        for (int i = 0; i < 384; ++i) r = g.add(r, r);
        g.r2_0 = r.lo;
        g.r2_1 = r.mid;
        g.r2_2 = r.hi;
        store_policy(c, g, POL_MONT192, PB_RED_MONT);
        return PB_OK;
    }
    if (!(mod[0] & 1)) return PB_EMODULUS;
    PM192 f;
    f.p0 = mod[0];
    f.p1 = mod[1];
    f.p2 = m2;
    f.mask_hi = top_mask;
    f.c = (uint32_t)cc;
    f.k = (uint32_t)k;
    store_policy(c, f, POL_PM192, PB_RED_PM);
    return PB_OK;
}

inline int build_binary_policy(PolicyBlob* c, const uint64_t* mod, int nlimbs) {
    uint64_t m0 = mod[0], m1 = nlimbs > 1 ? mod[1] : 0, m2 = nlimbs > 2 ? mod[2] : 0;
    int deg;
    if (m2) {
        if (m2 != 1) return PB_ENOTSUP;
        deg = 128;
    } else if (m1) {
        deg = 64 + (63 - __builtin_clzll(m1));
    } else if (m0) {
        deg = 63 - __builtin_clzll(m0);
    } else {
        return PB_EMODULUS;
    }
    if (deg < 1) return PB_EMODULUS;
    if (deg <= 8) {
        GF2P8 f;
        f.n = (uint32_t)deg;
        uint32_t r = (uint32_t)(m0 ^ (1ull << deg));
        f.red = r * 0x01010101u;
        f.top = (1u << (deg - 1)) * 0x01010101u;
        f.emask = ((1u << deg) - 1) * 0x01010101u;
        store_policy(c, f, POL_GF2P8, PB_RED_SWAR);
        return PB_OK;
    }
    if (deg <= 32) {
        GF2W32 f;
        f.n = (uint32_t)deg;
        f.emask = deg == 32 ? ~0u : ((1u << deg) - 1);
        f.red = (uint32_t)(m0 ^ (1ull << deg));
        f.fast = 0;
        if (f.red && f.red < (1u << 28)) {
            // fold passes until nothing sticks out above bit n: excess e -> max(0, e + deg(r) - n)
            int rdeg = 31 - __builtin_clz(f.red);
            int e = deg - 1, folds = 0;
            while (e > 0 && folds < 16) {
                e = e + rdeg - deg;
                if (e < 0) e = 0;
                ++folds;
            }
            // a pass costs ~2 + 2 popcount(red) instructions, a long-division step ~5: fold only when it is the cheaper one
            if (e == 0 && folds * (2 + 2 * __builtin_popcount(f.red)) < 5 * (deg - 1)) f.fast = 1u | ((uint32_t)folds << 8);
        }
        store_policy(c, f, POL_GF2W32, PB_RED_WIDE);
        return PB_OK;
    }
    if (deg <= 64) {
        GF2W64 f;
        f.n = (uint32_t)deg;
        f.emask = deg == 64 ? ~0ull : ((1ull << deg) - 1);
        f.red = deg == 64 ? m0 : (m0 ^ (1ull << deg));
        f.fast = 0;
        if (f.red && f.red < (1ull << 28)) {
            // fold passes until nothing sticks out above bit n: excess e -> max(0, e + deg(r) - n)
            int rdeg = 63 - __builtin_clzll(f.red);
            int e = deg - 1, folds = 0;
            while (e > 0 && folds < 16) {
                e = e + rdeg - deg;
                if (e < 0) e = 0;
                ++folds;
            }
            // n <= 32: a pass costs ~2 + 2 popcount(red) instructions, a long-division step ~5: fold only when it is the cheaper one
            const bool pays = deg > 32 || folds * (2 + 2 * __builtin_popcountll(f.red)) < 5 * (deg - 1);
            if (e == 0 && pays) f.fast = 1u | ((uint32_t)folds << 8);
        }
        store_policy(c, f, POL_GF2W64, PB_RED_WIDE);
        return PB_OK;
    }
    GF2W128 f;
    f.n = (uint32_t)deg;
    f.red_lo = m0;
    f.red_hi = deg == 128 ? m1 : (m1 ^ (1ull << (deg - 64)));
    f.emask_lo = ~0ull;
    f.emask_hi = deg == 128 ? ~0ull : ((1ull << (deg - 64)) - 1);
    f.fast = (f.red_hi == 0 && f.red_lo != 0 && f.red_lo < (1ull << 28)) ? 1u : 0u;
    store_policy(c, f, POL_GF2W128, PB_RED_WIDE);
    return PB_OK;
}



// 2^W mod p for the keystream sampler (rng.hpp): W = 32 / 64 / 128 by storage width.
template <class F>
inline void rng_const_prime(const PolicyBlob& pb, int W, uint64_t out[2]) {
    F f;
    memcpy(&f, pb.bytes, sizeof(F));
    typename F::word r;
    memset(&r, 0, sizeof(r));
    ((unsigned char*)&r)[0] = 1;  // little-endian 1
    for (int i = 0; i < W; ++i) r = f.add(r, r);
    out[0] = out[1] = 0;
    memcpy(out, &r, sizeof(r) < 16 ? sizeof(r) : 16);
}

inline void rng_const(const PolicyBlob& pb, uint64_t out[2]) {
    out[0] = out[1] = 0;
    switch (pb.kind) {
        case POL_PM64_MERSENNE: rng_const_prime<PM64<false, true> >(pb, 64, out); break;
        case POL_PM64_K64: rng_const_prime<PM64<true, false> >(pb, 64, out); break;
        case POL_PM64_GEN: rng_const_prime<PM64<false, false> >(pb, 64, out); break;
        case POL_RC64: rng_const_prime<RC64>(pb, 64, out); break;
        case POL_RC32: rng_const_prime<RC32>(pb, 32, out); break;
        case POL_PM128_K128: rng_const_prime<PM128<true> >(pb, 128, out); break;
        case POL_PM128_GEN: rng_const_prime<PM128<false> >(pb, 128, out); break;
        case POL_PM96: rng_const_prime<PM96>(pb, 128, out); break;
        case POL_MONT128: rng_const_prime<MONT128>(pb, 128, out); break;
        case POL_PM192: rng_const_prime<PM192>(pb, 192, out); break;      // c * 2^(192-k) < 2^94: two limbs
        default: break;  // binary fields: masks only
    }
}

}  // namespace ffgpu
