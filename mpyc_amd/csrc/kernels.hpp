// kernels.hpp -- HIP kernels of libffgpu, templated on a field policy (fields.hpp).
//
// Every kernel here is a streaming, HBM-bound integer kernel (no MFMA: there
// is no dense contraction on this path).  Design rules, from the gfx950 guide:
//   * 16 bytes per lane per access (global_load/store_dwordx4): a wave touches
//     1 KiB of contiguous HBM per instruction and array;
//   * all loads of a pack are issued before the first use, so a wave has
//     (rows) independent 1 KiB requests in flight and a CU up to 32 x that;
//   * one 16-byte pack per thread and an UNCAPPED grid (n/2/256 workgroups for
//     64-bit fields, ~19.5k for n = 10^7): at 40-50 us per launch the hardware
//     dispatcher balances the tail better than a capped grid-stride loop
//     (measured: profiles/r01_tuning.md).  Consecutive blocks (which land on
//     different XCDs, block b -> XCD b%8) touch consecutive 4 KiB chunks, so
//     every XCD's L2 and all HBM channels see the same uniform stream; there is
//     no reuse to localise in an L2, hence no XCD remap;
//   * streamed-once data carries the non-temporal hint (nt) on loads and stores;
//   * per-field constants (modulus, fold constant, Lagrange vector, party
//     x-coordinates) are wave-uniform kernel arguments -> SGPRs, which beats
//     staging them in LDS for fields this small.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <type_traits>
#include <utility>
#include "fields.hpp"
#include "rng.hpp"

namespace ffgpu {

enum { BLOCK = 256 };
enum EwOp { OP_ADD = 0, OP_SUB = 1, OP_MUL = 2, OP_RSUB = 3, OP_NEG = 4, OP_REDUCE = 5, OP_COPY = 6 };

enum { MAXK = 9, MAXW = 8, MAXK_ANY = 64, MAXT = 4 };

template <class W>
struct alignas(16) Pack {
    enum { N = 16 / sizeof(W) };
    W w[N];
};
template <>
struct Pack<u192e> {          // three-limb elements: one per lane (24 bytes: dwordx4 + dwordx2)
    enum { N = 1 };
    u192e w[1];
};

// 16-byte global accesses with an optional non-temporal hint.  Every array here is
// streamed exactly once per launch, so by default loads and stores carry `nt`
// (global_load/store_dwordx4 ... nt): measured +4..10 % on the 10^7-element kernels
// (profiles/r01_tuning.md); every launcher instantiates the hinted form only.
typedef uint32_t ff_u32x4 __attribute__((ext_vector_type(4)));

template <bool NT, class P>
__device__ __forceinline__ P ldg(const P* p) {
    static_assert(sizeof(P) == 16, "16-byte packs only");
    if constexpr (NT) {
        ff_u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const ff_u32x4*>(p));
        P r;
        __builtin_memcpy(&r, &v, 16);
        return r;
    } else {
        return *p;
    }
}
template <bool NT, class P>
__device__ __forceinline__ void stg(P* p, const P& x) {
    static_assert(sizeof(P) == 16, "16-byte packs only");
    if constexpr (NT) {
        ff_u32x4 v;
        __builtin_memcpy(&v, &x, 16);
        __builtin_nontemporal_store(v, reinterpret_cast<ff_u32x4*>(p));
    } else {
        *p = x;
    }
}

// Memory-side pack of a field: what one lane moves per access.  16 bytes (= the register pack) for all
// fields but PM96, whose 12-byte elements go one per lane through dwordx3 accesses.
template <class F>
struct MemPack {
    typedef Pack<typename F::word> type;
};
template <>
struct MemPack<PM96> {
    typedef e96 type;
};
template <>
struct MemPack<PM192> {
    typedef u192e type;
};
template <>
struct MemPack<MONT192> {
    typedef u192e type;
};
typedef uint32_t ff_u32x2 __attribute__((ext_vector_type(2)));

template <bool NT>
__device__ __forceinline__ Pack<u192e> ldg(const u192e* p) {
    const ff_u32x2* q = reinterpret_cast<const ff_u32x2*>(p);     // 8-byte aligned elements: dwordx2 x 3
    ff_u32x2 a0, a1, b;
    if constexpr (NT) {
        a0 = __builtin_nontemporal_load(q);
        a1 = __builtin_nontemporal_load(q + 1);
        b = __builtin_nontemporal_load(q + 2);
    } else {
        a0 = q[0];
        a1 = q[1];
        b = q[2];
    }
    Pack<u192e> r;
    r.w[0].lo = (uint64_t)a0.x | ((uint64_t)a0.y << 32);
    r.w[0].mid = (uint64_t)a1.x | ((uint64_t)a1.y << 32);
    r.w[0].hi = (uint64_t)b.x | ((uint64_t)b.y << 32);
    return r;
}
template <bool NT>
__device__ __forceinline__ void stg(u192e* p, const Pack<u192e>& x) {
    ff_u32x2* q = reinterpret_cast<ff_u32x2*>(p);
    ff_u32x2 a0, a1, b;
    a0.x = (uint32_t)x.w[0].lo;  a0.y = (uint32_t)(x.w[0].lo >> 32);
    a1.x = (uint32_t)x.w[0].mid; a1.y = (uint32_t)(x.w[0].mid >> 32);
    b.x = (uint32_t)x.w[0].hi;   b.y = (uint32_t)(x.w[0].hi >> 32);
    if constexpr (NT) {
        __builtin_nontemporal_store(a0, q);
        __builtin_nontemporal_store(a1, q + 1);
        __builtin_nontemporal_store(b, q + 2);
    } else {
        q[0] = a0;
        q[1] = a1;
        q[2] = b;
    }
}
typedef uint32_t ff_u32x3 __attribute__((ext_vector_type(3)));
template <bool NT>
__device__ __forceinline__ Pack<u128e> ldg(const e96* p) {
    ff_u32x3 v;
    if constexpr (NT) v = __builtin_nontemporal_load(reinterpret_cast<const ff_u32x3*>(p));
    else v = *reinterpret_cast<const ff_u32x3*>(p);
    Pack<u128e> r;
    r.w[0].lo = (uint64_t)v.x | ((uint64_t)v.y << 32);
    r.w[0].hi = v.z;
    return r;
}
template <bool NT>
__device__ __forceinline__ void stg(e96* p, const Pack<u128e>& x) {
    ff_u32x3 v;
    v.x = (uint32_t)x.w[0].lo;
    v.y = (uint32_t)(x.w[0].lo >> 32);
    v.z = (uint32_t)x.w[0].hi;
    if constexpr (NT) __builtin_nontemporal_store(v, reinterpret_cast<ff_u32x3*>(p));
    else *reinterpret_cast<ff_u32x3*>(p) = v;
}

// ---- wave-contiguous accesses: ldgw / stgw ---------------------------------------------------------------------------
// CONTRACT (the streaming loops `for (i = gid; i < nvec; i += gsz)` of this file, with the launchers' nvec): all 64 lanes of
// the wave are active and lane L accesses pack (first lane's pack) + L.  For every pack type but the 24-byte one this is
// ldg / stg.  24-byte elements (three-limb primes) at one element per lane have a 24-byte lane stride: as three dwordx2 per
// lane every wave instruction touches twelve 128-byte lines and uses a third of each -- 5.2-5.4 TB/s where the 16-byte
// fields stream at 6.2-6.4 (tools/tune_x24.hip).  Round 6: the WAVE moves its 64 elements = 1536 contiguous bytes as 96
// dwordx4 accesses (one instruction on all lanes, one on lanes 0..31) and the lanes pick their own 24 bytes out of a
// per-wave LDS region (written 16 bytes per lane, read back as three 8-byte words at a 24-byte stride, or the other way
// round: conflict-free either way); no barrier: a wave's LDS instructions execute in order, a wavefront-scope fence on
// the LDS address space alone keeps the compiler from reordering them.
// next iteration's arrays are not held up by anything else.  The launchers (launch.hpp, Launchers<F>::nvec_of) hand 24-byte fields
// whole waves only (nvec a multiple of 64, the rest goes through the scalar tail) and require 16-byte aligned rows.
// Loads come in two steps so that a kernel puts ALL its global loads in flight before the first LDS round trip:   auto ra = ldgw_issue<NT>(pa), rb = ldgw_issue<NT>(pb);
//                                                  P x = ldgw_finish(ra), y = ldgw_finish(rb);
template <bool NT, class P>
__device__ __forceinline__ auto ldgw_issue(const P* p) { return ldg<NT>(p); }
template <class X>
__device__ __forceinline__ X ldgw_finish(const X& x) { return x; }
template <bool NT, class P, class X>
__device__ __forceinline__ void stgw(P* p, const X& x) { stg<NT>(p, x); }

typedef __attribute__((address_space(3))) ff_u32x4 x24_lds4;
typedef __attribute__((address_space(3))) ff_u32x2 x24_lds2;
__device__ __forceinline__ x24_lds4* x24_region() {
    __shared__ __attribute__((aligned(16))) ff_u32x4 x24_lds[BLOCK / 64][96];
    return (x24_lds4*)(ff_u32x4*)x24_lds[threadIdx.x >> 6];
}
// orders the wave's LDS accesses (and only those) for the compiler; emits no instruction -- a wave's LDS instructions execute
// in order, which is all the lanes need to see each other's writes
#define X24_LDS_FENCE() __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront", "local")
// first byte of the wave's 64 elements, from the lane's own pointer (the contract: lane L holds first + L): pointer
// arithmetic on the kernel argument keeps the global address space (global_load, not flat_load)
__device__ __forceinline__ const ff_u32x4* x24_wave_base(const u192e* p) {
    return reinterpret_cast<const ff_u32x4*>(reinterpret_cast<const char*>(p) - (size_t)__lane_id() * 24);
}
__device__ __forceinline__ ff_u32x4* x24_wave_base(u192e* p) {
    return reinterpret_cast<ff_u32x4*>(reinterpret_cast<char*>(p) - (size_t)__lane_id() * 24);
}
// A wave's 1536 bytes in flight: chunk `lane` and chunk 64 + (lane & 31).  Lanes 32..63 repeat the second access of lanes
// 0..31 (same address; for stores the same data too) instead of sitting it out under an execution mask: a masked access is
// a branch, and every branch between the loads of a kernel is a point where the compiler waits for ALL of them.
struct X24Raw {
    ff_u32x4 r0, r1;
};
template <bool NT>
__device__ __forceinline__ X24Raw ldgw_issue(const u192e* p) {
    const ff_u32x4* base = x24_wave_base(p);
    const uint32_t lane = __lane_id();
    X24Raw r;
    if constexpr (NT) {
        r.r0 = __builtin_nontemporal_load(base + lane);
        r.r1 = __builtin_nontemporal_load(base + 64 + (lane & 31));
    } else {
        r.r0 = base[lane];
        r.r1 = base[64 + (lane & 31)];
    }
    return r;
}
__device__ __forceinline__ Pack<u192e> ldgw_finish(const X24Raw& raw) {
    const uint32_t lane = __lane_id();
    x24_lds4* L = x24_region();
    X24_LDS_FENCE();                                       // (after the reads of the previous round trip)
    L[lane] = raw.r0;
    L[64 + (lane & 31)] = raw.r1;
    X24_LDS_FENCE();
    const x24_lds2* R = reinterpret_cast<const x24_lds2*>(L) + 3 * lane;
    const ff_u32x2 a0 = R[0], a1 = R[1], b = R[2];
    Pack<u192e> r;
    r.w[0].lo = (uint64_t)a0.x | ((uint64_t)a0.y << 32);
    r.w[0].mid = (uint64_t)a1.x | ((uint64_t)a1.y << 32);
    r.w[0].hi = (uint64_t)b.x | ((uint64_t)b.y << 32);
    return r;
}
template <bool NT>
__device__ __forceinline__ void stgw(u192e* p, const Pack<u192e>& x) {
    ff_u32x4* base = x24_wave_base(p);
    const uint32_t lane = __lane_id();
    x24_lds4* L = x24_region();
    x24_lds2* R = reinterpret_cast<x24_lds2*>(L) + 3 * lane;
    ff_u32x2 a0, a1, b;
    a0.x = (uint32_t)x.w[0].lo;  a0.y = (uint32_t)(x.w[0].lo >> 32);
    a1.x = (uint32_t)x.w[0].mid; a1.y = (uint32_t)(x.w[0].mid >> 32);
    b.x = (uint32_t)x.w[0].hi;   b.y = (uint32_t)(x.w[0].hi >> 32);
    X24_LDS_FENCE();
    R[0] = a0;
    R[1] = a1;
    R[2] = b;
    X24_LDS_FENCE();
    const ff_u32x4 r0 = L[lane], r1 = L[64 + (lane & 31)];
    if constexpr (NT) {
        __builtin_nontemporal_store(r0, base + lane);
        __builtin_nontemporal_store(r1, base + 64 + (lane & 31));
    } else {
        base[lane] = r0;
        base[64 + (lane & 31)] = r1;
    }
}

// element <-> word for the scalar tail (identity unless words pack elements)
template <class F>
__device__ __forceinline__ typename F::word ld_elem(const typename F::elem* p, size_t i) {
    if constexpr (sizeof(typename F::elem) == 12) {
        typename F::word w;
        w.lo = (uint64_t)p[i].x[0] | ((uint64_t)p[i].x[1] << 32);
        w.hi = p[i].x[2];
        return w;
    } else if constexpr (F::EPW == 1) {
        return p[i];
    } else {
        return (typename F::word)p[i];
    }
}
template <class F>
__device__ __forceinline__ void st_elem(typename F::elem* p, size_t i, typename F::word w) {
    if constexpr (sizeof(typename F::elem) == 12) {
        p[i].x[0] = (uint32_t)w.lo;
        p[i].x[1] = (uint32_t)(w.lo >> 32);
        p[i].x[2] = (uint32_t)w.hi;
    } else if constexpr (F::EPW == 1) {
        p[i] = w;
    } else {
        p[i] = (typename F::elem)w;
    }
}

template <class F, int OP>
__device__ __forceinline__ typename F::word ew_apply(const F& f, typename F::word a, typename F::word b) {
    if constexpr (OP == OP_ADD) return f.add(a, b);
    else if constexpr (OP == OP_SUB) return f.sub(a, b);
    else if constexpr (OP == OP_MUL) return f.mul(a, b);
    else if constexpr (OP == OP_RSUB) return f.sub(b, a);
    else if constexpr (OP == OP_NEG) return f.neg(a);
    else if constexpr (OP == OP_REDUCE) return f.reduce_raw(a);
    else return a;
}

// ---- out = a (op) b --------------------------------------------------------
template <class F, int OP, bool NT>
__device__ __forceinline__ void ew2_body(const F& f, const typename F::elem* __restrict__ a, const typename F::elem* __restrict__ b,
                                         typename F::elem* __restrict__ o, size_t nvec, size_t n) {
    typedef Pack<typename F::word> P;
    typedef typename MemPack<F>::type MP;
    const MP* __restrict__ av = reinterpret_cast<const MP*>(a);
    const MP* __restrict__ bv = reinterpret_cast<const MP*>(b);
    MP* __restrict__ ov = reinterpret_cast<MP*>(o);
    const size_t gid = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    const size_t gsz = (size_t)gridDim.x * BLOCK;
    for (size_t i = gid; i < nvec; i += gsz) {
        const auto rx = ldgw_issue<NT>(av + i);
        const auto ry = ldgw_issue<NT>(bv + i);
        P x = ldgw_finish(rx);
        P y = ldgw_finish(ry);
        P r;
#pragma unroll
        for (int q = 0; q < P::N; ++q) r.w[q] = ew_apply<F, OP>(f, x.w[q], y.w[q]);
        stgw<NT>(ov + i, r);
    }
    // scalar tail (n not a multiple of the pack size, or unaligned pointers: nvec == 0)
    const size_t done = nvec * (size_t)(P::N * F::EPW);
    for (size_t e = done + gid; e < n; e += gsz) {
        st_elem<F>(o, e, ew_apply<F, OP>(f, ld_elem<F>(a, e), ld_elem<F>(b, e)));
    }
}
template <class F, int OP, bool NT>
__global__ __launch_bounds__(BLOCK) void k_ew2(F f, const typename F::elem* __restrict__ a,
                                                const typename F::elem* __restrict__ b,
                                                typename F::elem* __restrict__ o, size_t nvec, size_t n) {
    ew2_body<F, OP, NT>(f, a, b, o, nvec, n);
}
// The same kernel held to WAVES waves per SIMD, for products that the VALU limits: left alone the compiler hoists all nine
// 32 x 32 carry-less products of a GF(2^128) multiplication side by side (129 registers, three waves per SIMD); told to keep
// five or more waves it needs 66 registers and no spill, and seven waves issue 4-5 % faster than three (round 6, same box:
// 158-160 -> 151-153 us at n = 10^7).  EwOccupancy<F, OP>::waves names the kernels treated this way.
template <class F, int OP, bool NT, int WAVES>
__global__ __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu(WAVES, 8)))
void k_ew2_occ(F f, const typename F::elem* __restrict__ a, const typename F::elem* __restrict__ b,
               typename F::elem* __restrict__ o, size_t nvec, size_t n) {
    ew2_body<F, OP, NT>(f, a, b, o, nvec, n);
}
template <class F, int OP> struct EwOccupancy { enum { waves = 0 }; };
#ifndef FFGPU_GF2W128_OCC
#define FFGPU_GF2W128_OCC 7
#endif
template <> struct EwOccupancy<GF2W128, OP_MUL> { enum { waves = FFGPU_GF2W128_OCC }; };

// ---- out = a (op) scalar, or unary op (scalar ignored) ---------------------
template <class F, int OP, bool NT>
__global__ __launch_bounds__(BLOCK) void k_ew1(F f, const typename F::elem* __restrict__ a,
                                                typename F::word s, typename F::elem* __restrict__ o,
                                                size_t nvec, size_t n) {
    typedef Pack<typename F::word> P;
    typedef typename MemPack<F>::type MP;
    const MP* __restrict__ av = reinterpret_cast<const MP*>(a);
    MP* __restrict__ ov = reinterpret_cast<MP*>(o);
    const size_t gid = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    const size_t gsz = (size_t)gridDim.x * BLOCK;
    for (size_t i = gid; i < nvec; i += gsz) {
        P x = ldgw_finish(ldgw_issue<NT>(av + i));
        P r;
#pragma unroll
        for (int q = 0; q < P::N; ++q) r.w[q] = ew_apply<F, OP>(f, x.w[q], s);
        stgw<NT>(ov + i, r);
    }
    const size_t done = nvec * (size_t)(P::N * F::EPW);
    for (size_t e = done + gid; e < n; e += gsz) {
        st_elem<F>(o, e, ew_apply<F, OP>(f, ld_elem<F>(a, e), s));
    }
}

// ---- out = a*b + c ---------------------------------------------------------
template <class F, bool NT>
__global__ __launch_bounds__(BLOCK) void k_muladd(F f, const typename F::elem* __restrict__ a,
                                                   const typename F::elem* __restrict__ b,
                                                   const typename F::elem* __restrict__ c,
                                                   typename F::elem* __restrict__ o, size_t nvec, size_t n) {
    typedef Pack<typename F::word> P;
    typedef typename MemPack<F>::type MP;
    const MP* __restrict__ av = reinterpret_cast<const MP*>(a);
    const MP* __restrict__ bv = reinterpret_cast<const MP*>(b);
    const MP* __restrict__ cv = reinterpret_cast<const MP*>(c);
    MP* __restrict__ ov = reinterpret_cast<MP*>(o);
    const size_t gid = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    const size_t gsz = (size_t)gridDim.x * BLOCK;
    for (size_t i = gid; i < nvec; i += gsz) {
        const auto rx = ldgw_issue<NT>(av + i);
        const auto ry = ldgw_issue<NT>(bv + i);
        const auto rz = ldgw_issue<NT>(cv + i);
        P x = ldgw_finish(rx);
        P y = ldgw_finish(ry);
        P z = ldgw_finish(rz);
        P r;
#pragma unroll
        for (int q = 0; q < P::N; ++q) r.w[q] = f.muladd(x.w[q], y.w[q], z.w[q]);
        stgw<NT>(ov + i, r);
    }
    const size_t done = nvec * (size_t)(P::N * F::EPW);
    for (size_t e = done + gid; e < n; e += gsz) {
        st_elem<F>(o, e, f.muladd(ld_elem<F>(a, e), ld_elem<F>(b, e), ld_elem<F>(c, e)));
    }
}

// ---- Beaver-triple combination: out = z + d*y + e*x + d*e ---------------------------------------
// (d = a - x and e = b - y are the opened masked operands, [x],[y],[z = xy] the triple shares.)
// NOT a reference function: MPyC multiplies with GRR resharing (runtime.py:603-689), it has no Beaver
// triples.  Provided because the project brief names it; parity for this entry point is UNPINNED --
// it is checked only against the textbook identity (tests/test_gpu_parity.py::test_beaver_combine).
template <class F, bool NT>
__global__ __launch_bounds__(BLOCK) void k_beaver(F f, const typename F::elem* __restrict__ z,
                                                   const typename F::elem* __restrict__ x,
                                                   const typename F::elem* __restrict__ y,
                                                   const typename F::elem* __restrict__ d,
                                                   const typename F::elem* __restrict__ e,
                                                   typename F::elem* __restrict__ o, int add_de, size_t nvec, size_t n) {
    typedef Pack<typename F::word> P;
    typedef typename MemPack<F>::type MP;
    typedef typename F::word W;
    const size_t gid = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    const size_t gsz = (size_t)gridDim.x * BLOCK;
    auto comb = [&](W zz, W xx, W yy, W dd, W ee) -> W {
        W c = f.muladd(dd, yy, zz);
        c = f.muladd(ee, xx, c);
        return add_de ? f.muladd(dd, ee, c) : c;       // public term: all parties (Shamir) / one party (additive)
    };
    for (size_t i = gid; i < nvec; i += gsz) {
        const auto rz = ldgw_issue<NT>(reinterpret_cast<const MP*>(z) + i), rx = ldgw_issue<NT>(reinterpret_cast<const MP*>(x) + i);
        const auto ry = ldgw_issue<NT>(reinterpret_cast<const MP*>(y) + i), rd = ldgw_issue<NT>(reinterpret_cast<const MP*>(d) + i);
        const auto re = ldgw_issue<NT>(reinterpret_cast<const MP*>(e) + i);
        P pz = ldgw_finish(rz), px = ldgw_finish(rx), py = ldgw_finish(ry), pd = ldgw_finish(rd), pe = ldgw_finish(re), r;
#pragma unroll
        for (int q = 0; q < P::N; ++q) r.w[q] = comb(pz.w[q], px.w[q], py.w[q], pd.w[q], pe.w[q]);
        stgw<NT>(reinterpret_cast<MP*>(o) + i, r);
    }
    const size_t done = nvec * (size_t)(P::N * F::EPW);
    for (size_t k_ = done + gid; k_ < n; k_ += gsz)
        st_elem<F>(o, k_, comb(ld_elem<F>(z, k_), ld_elem<F>(x, k_), ld_elem<F>(y, k_), ld_elem<F>(d, k_), ld_elem<F>(e, k_)));
}

// ---- Shamir share generation (thresha.py:47-64), optionally fused with the
//      local product of secure multiplication (runtime.py:1134) ---------------
// share_i[h] = s[h] + x_i*(C[0][h] + x_i*(C[1][h] + ... x_i*C[T-1][h])),  x_i = i+1
// Per pack: 1 (or 2) + T loads of 16 B, m stores of 16 B, m*T Horner steps.
struct RngArgs {
    RngKey rk;
    uint64_t r0, r1;  // 2^W mod p for the sampler
    // device-resident generator state (ffgpu_rng_state_*): when set, key / nonce / rounds are read from it at
    // kernel start instead of from the kernel arguments, so a captured HIP graph draws fresh coefficients on
    // every replay (k_rng_bump advances the nonce after each use)
    const RngKey* dev_key;
    int spread;   // 1: one pack per thread (small arrays), each thread recomputing its group's keystream
    // GF(2^n<=8) fused local product: log/antilog tables in device memory (512 B of u16 logs, then 1024 B of
    // antilogs; misc.hip Gf8Tables).  When set, the product of the fused kernel goes through LDS lookups,
    // which run beside the ChaCha VALU work instead of adding ~110 VALU ops per word to it.
    const void* aux;
    uint32_t nonce_off;  // device-resident state: added to the state's nonce for THIS launch (deferred advance: a
                         // sequence of launches uses offsets 0, 1, 2, ... and ONE ffgpu_rng_state_advance follows)
    int no_advance;      // 1: this launch leaves the device-resident nonce alone (the caller advances it)
    int release;  // 1: the kernel's last workgroup advances the device-resident nonce (small grids only: one
                  // atomic per workgroup on one address serialises, ~25 ns each); 0: k_rng_advance follows
};


// ---- operands of a gate given as RECOMBINATIONS (the fused chain kernel) ---------------------------------
// In a chain of secure multiplications a party's new share y = sum_j lambda_j r_j (r_j = the sub-shares it
// received, thresha.py:119-132) is consumed by the local product of the next gate (runtime.py:1134).  With
// REC the share-generation kernel takes both factors in that form -- k rows and their Lagrange vector each; a
// factor that already exists as an array is the 1-row case with lambda = 1 -- recombines them in registers,
// multiplies and re-shares: y never goes to HBM and back (6 instead of 9 accesses per element for a squaring).
enum { GATE_MAXK = 7 };
template <class F>
struct GateSrc {
    const typename F::elem* rowsA[GATE_MAXK];
    const typename F::elem* rowsB[GATE_MAXK];
    typename F::word lamA[GATE_MAXK], lamB[GATE_MAXK];   // prepared (f.prep)
    int kA, kB;
    int square;                                          // 1: second factor = first factor
    int plainA, plainB;                                  // 1: a single row with lambda = 1 (an existing array)
    // batched launch (gridDim.y senders in one grid, e.g. all parties of a computation held on one GPU): workgroup
    // row y reads every row of A at element offset y*yA (B: y*yB) and writes its share rows at offset y*yO; its
    // generator stream is the call's with y added to bits 8..15 of nonce word 1
    size_t yA, yB, yO;
};

// Policies with a dot product in 28-bit digits (fields.hpp LazyDot: the multi-limb 2^k - c primes) take it for sums of at
// most FF_D28_MAX_TERMS terms -- the recombination kernels' K <= MAXK rows; everything else accumulates in F::acc.
template <class F, class = void>
struct HasLazyAcc : std::false_type {};
template <class F>
struct HasLazyAcc<F, std::void_t<typename F::lacc> > : std::true_type {};
static_assert(MAXK <= FF_D28_MAX_TERMS, "the digit accumulator's column bound");
template <class F, class = void>
struct LazyAccOf {
    typedef typename F::acc type;
};
template <class F>
struct LazyAccOf<F, std::void_t<typename F::lacc> > {
    typedef typename F::lacc type;
};

template <class F, bool NT>
__device__ __forceinline__ Pack<typename F::word> gate_load(const F& f, const typename F::elem* const* rows,
                                                            const typename F::word* lam, int k, size_t i) {
    typedef Pack<typename F::word> P;
    typedef typename MemPack<F>::type MP;
    static_assert(GATE_MAXK <= FF_D28_MAX_TERMS, "the digit accumulator's column bound");
    using Acc = typename std::conditional<HasLazyAcc<F>::value, typename LazyAccOf<F>::type, typename F::acc>::type;
    Acc acc[P::N];
#pragma unroll
    for (int q = 0; q < P::N; ++q) {
        if constexpr (HasLazyAcc<F>::value) f.lacc_zero(acc[q]); else f.acc_zero(acc[q]);
    }
    // rows in chunks of four: the loads of a chunk are issued together and waited for once (k is wave-uniform, the
    // guards are scalar branches) -- one load, one wait, one multiply-add per row would expose k memory latencies
    for (int j0 = 0; j0 < k; j0 += 4) {
        P x[4];
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (j0 + u < k) x[u] = ldg<NT>(reinterpret_cast<const MP*>(rows[j0 + u]) + i);
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (j0 + u < k) {
#pragma unroll
                for (int q = 0; q < P::N; ++q) {
                    if constexpr (HasLazyAcc<F>::value) f.lacc_mac(acc[q], lam[j0 + u], x[u].w[q]);
                    else f.acc_mac(acc[q], lam[j0 + u], x[u].w[q]);
                }
            }
    }
    P r;
#pragma unroll
    for (int q = 0; q < P::N; ++q) {
        if constexpr (HasLazyAcc<F>::value) r.w[q] = f.lacc_reduce(acc[q]); else r.w[q] = f.acc_reduce(acc[q]);
    }
    return r;
}
template <class F>
__device__ __forceinline__ typename F::word gate_load_elem(const F& f, const typename F::elem* const* rows,
                                                           const typename F::word* lam, int k, size_t e) {
    typename F::acc acc;
    f.acc_zero(acc);
    for (int j = 0; j < k; ++j) {
        f.acc_mac(acc, lam[j], ld_elem<F>(rows[j], e));     // packed fields: the element sits in the low byte of the word
    }
    return f.acc_reduce(acc);
}

// Device-resident generator state: the LAST workgroup to finish advances the nonce (every workgroup has read
// the state by then; the next launch on the stream starts after this one ends).  pad_ counts finished groups.
__device__ __forceinline__ void rng_state_release(const RngArgs& ra) {
    if (!ra.dev_key || !ra.release) return;
    __syncthreads();
    if (threadIdx.x == 0) {
        RngKey* st = const_cast<RngKey*>(ra.dev_key);
        __threadfence();
        const uint32_t done = atomicAdd(&st->pad_, 1u);
        if (done == gridDim.x * gridDim.y - 1) {
            st->pad_ = 0;
            if (++st->nonce[0] == 0) st->nonce[1] += 65536u;   // rows of t > 4 calls use nonce[1] + j + 1, j < 64
            __threadfence();
        }
    }
}

// large grids: a one-thread kernel after the share-generation kernel (stream order) advances the nonce
template <int UNUSED>
__global__ void k_rng_advance(RngKey* st, uint32_t by) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        const uint32_t old = st->nonce[0];
        st->nonce[0] = old + by;
        if (st->nonce[0] < old) st->nonce[1] += 65536u;
    }
}
// kernel start: key / nonce / rounds from the device-resident state, plus this launch's offset
__device__ __forceinline__ void rng_load_state(RngArgs& ra) {
    if (ra.dev_key) {
        ra.rk = *ra.dev_key;
        const uint32_t old = ra.rk.nonce[0];
        ra.rk.nonce[0] = old + ra.nonce_off;
        if (ra.rk.nonce[0] < old) ra.rk.nonce[1] += 65536u;
    }
}
enum { RNG_RELEASE_MAX_GRID = 512 };

template <class F, int T, bool FUSE_MUL, bool NT, bool RNG, bool REC = false>
__global__ __launch_bounds__(BLOCK) void k_split(F f, const typename F::elem* __restrict__ a,
                                                  const typename F::elem* __restrict__ b,
                                                  const typename F::elem* __restrict__ coef, size_t cstride,
                                                  int m, typename F::elem* __restrict__ out, size_t ostride,
                                                  size_t nvec, size_t n, RngArgs ra, GateSrc<F> gs) {
    rng_load_state(ra);
    // batched launch: gate y = blockIdx.y reads its operand rows yA / yB elements further on and writes yO further on.
    // (The offsets are added where the rows are indexed: the row-pointer arrays stay read-only kernel arguments --
    // a modified copy would be demoted to scratch memory, since they are indexed by the runtime row count.)
    size_t yoffA = 0, yoffB = 0;
    if constexpr (REC) {
        const size_t yb = blockIdx.y;                  // wave-uniform
        yoffA = yb * gs.yA;
        yoffB = yb * gs.yB;
        out += yb * gs.yO;
        ra.rk.nonce[1] += (uint32_t)yb << 8;
    }
    typedef Pack<typename F::word> P;
    typedef typename MemPack<F>::type MP;
    typedef typename F::word W;
    constexpr int TT = T > 0 ? T : 1;
    // wave-contiguous accesses (ldgw / stgw): the plain one-pack-per-thread loop only -- with the in-kernel generator a
    // thread serves packs u * ngroups + ig (waves are neither whole nor aligned there), the chain gate adds row offsets
    constexpr bool WC = !RNG && !REC;
    const size_t gid = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    const size_t gsz = (size_t)gridDim.x * BLOCK;
    const MP* __restrict__ av = reinterpret_cast<const MP*>(a);
    const MP* __restrict__ bv = reinterpret_cast<const MP*>(b);
    // With the in-kernel CSPRNG a thread serves a GROUP of G adjacent packs from shared keystream
    // blocks (rng.hpp RngLayout); without it G = 1 and this is the plain one-pack-per-thread loop.
    constexpr int G = (RNG && T > 0) ? RngLayout<F, TT, P::N>::G : 1;
    constexpr int EPV_ = P::N * F::EPW;
    const size_t npacks_all = (n + EPV_ - 1) / EPV_;               // the layout is defined over ALL packs of n
    const size_t ngroups = (RNG && T > 0) ? (npacks_all + G - 1) / G : nvec;
    constexpr bool TABMUL = (F::EPW == 4) && FUSE_MUL && RNG;
    __shared__ uint16_t s_lg[TABMUL ? 256 : 1];
    __shared__ uint32_t s_ex[TABMUL ? 256 : 1];
    if constexpr (TABMUL) {
        if (ra.aux) {
            s_lg[threadIdx.x] = reinterpret_cast<const uint16_t*>(ra.aux)[threadIdx.x];          // BLOCK == 256
            s_ex[threadIdx.x] = reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(ra.aux) + 512)[threadIdx.x];
            __syncthreads();
        }
    }
    decltype(ldgw_issue<NT>(av)) rc[TT];                     // coefficient rows in flight (wave-contiguous path)
    // one pack: loads and the optional local product ...
    auto load_s = [&](size_t i) -> P {
        P s, s2;
        if constexpr (REC) {
            const size_t iA = i + yoffA / EPV_, iB = i + yoffB / EPV_;      // batch offsets are whole packs on this path
            s = gs.plainA ? ldg<NT>(reinterpret_cast<const MP*>(gs.rowsA[0]) + iA)
                          : gate_load<F, NT>(f, gs.rowsA, gs.lamA, gs.kA, iA);
            s2 = gs.square ? s
                           : (gs.plainB ? ldg<NT>(reinterpret_cast<const MP*>(gs.rowsB[0]) + iB)
                                        : gate_load<F, NT>(f, gs.rowsB, gs.lamB, gs.kB, iB));
        } else {
            if constexpr (WC) {
                // every load of the pack -- operands AND coefficient rows -- is in flight before the first is consumed
                const auto rs = ldgw_issue<NT>(av + i);
                auto rs2 = rs;
                if constexpr (FUSE_MUL) rs2 = ldgw_issue<NT>(bv + i);
#pragma unroll
                for (int j = 0; j < T; ++j) rc[j] = ldgw_issue<NT>(reinterpret_cast<const MP*>(coef + (size_t)j * cstride) + i);
                s = ldgw_finish(rs);
                if constexpr (FUSE_MUL) s2 = ldgw_finish(rs2);
            } else {
                s = ldg<NT>(av + i);
                if constexpr (FUSE_MUL) s2 = ldg<NT>(bv + i);
            }
        }
        if constexpr (TABMUL) {
            if (ra.aux) {
                const uint8_t* ex = reinterpret_cast<const uint8_t*>(s_ex);
#pragma unroll
                for (int q = 0; q < P::N; ++q) {
                    uint32_t acc = 0;
#pragma unroll
                    for (int k8 = 0; k8 < 4; ++k8) {
                        const uint32_t lsum = (uint32_t)s_lg[((uint32_t)s.w[q] >> (8 * k8)) & 0xffu] +
                                              (uint32_t)s_lg[((uint32_t)s2.w[q] >> (8 * k8)) & 0xffu];
                        acc |= (uint32_t)ex[lsum] << (8 * k8);
                    }
                    s.w[q] = (W)acc;
                }
            } else {
#pragma unroll
                for (int q = 0; q < P::N; ++q) s.w[q] = f.mul(s.w[q], s2.w[q]);
            }
        } else if constexpr (FUSE_MUL) {
#pragma unroll
            for (int q = 0; q < P::N; ++q) s.w[q] = f.mul(s.w[q], s2.w[q]);
        }
        return s;
    };
    // ... then m share evaluations and m stores
    auto eval_store = [&](size_t i, const P& s, W (&c)[TT][P::N]) {
        if constexpr (!(RNG && T > 0)) {
            if constexpr (WC) {
#pragma unroll
                for (int j = 0; j < T; ++j) {                        // (issued by load_s)
                    const P t_ = ldgw_finish(rc[j]);
#pragma unroll
                    for (int q = 0; q < P::N; ++q) c[j][q] = t_.w[q];
                }
            } else {
#pragma unroll
                for (int j = 0; j < T; ++j) {
                    const P t_ = ldg<NT>(reinterpret_cast<const MP*>(coef + (size_t)j * cstride) + i);
#pragma unroll
                    for (int q = 0; q < P::N; ++q) c[j][q] = t_.w[q];
                }
            }
        }
        if constexpr (!F::BINARY && T >= 1) {
            // prime fields: forward differences over the consecutive party points (fields.hpp share_diff_*): T modular
            // additions per share and no multiplication (Horner: T multiply-adds by the point, ~4x the instructions --
            // what the kernels with the in-register ChaCha draw are bound by)
            W dd[P::N][TT];
#pragma unroll
            for (int q = 0; q < P::N; ++q) {
                W cq[TT];
#pragma unroll
                for (int j = 0; j < T; ++j) cq[j] = c[j][q];
                share_diff_init<F, TT>(f, cq, dd[q]);
            }
            P y = s;
            for (int party = 1; party <= m; ++party) {
#pragma unroll
                for (int q = 0; q < P::N; ++q) y.w[q] = share_diff_next<F, TT>(f, y.w[q], dd[q]);
                if constexpr (WC) stgw<NT>(reinterpret_cast<MP*>(out + (size_t)(party - 1) * ostride) + i, y);
                else stg<NT>(reinterpret_cast<MP*>(out + (size_t)(party - 1) * ostride) + i, y);
            }
        } else {
            // GF(2^n) (the points are field elements, not integers) and T = 0: Horner by the point
            for (int party = 1; party <= m; ++party) {
                P y;
                if constexpr (T == 0) {
                    y = s;
                } else {
#pragma unroll
                    for (int q = 0; q < P::N; ++q) {
                        W acc = c[T - 1][q];
#pragma unroll
                        for (int j = T - 2; j >= 0; --j) acc = f.muladd_small(acc, (uint32_t)party, c[j][q]);
                        y.w[q] = f.muladd_small(acc, (uint32_t)party, s.w[q]);
                    }
                }
                if constexpr (WC) stgw<NT>(reinterpret_cast<MP*>(out + (size_t)(party - 1) * ostride) + i, y);
                else stg<NT>(reinterpret_cast<MP*>(out + (size_t)(party - 1) * ostride) + i, y);
            }
        }
    };
    if (RNG && T > 0 && G > 1 && ra.spread) {
        // small arrays: one pack per thread, every thread of a group recomputes the group's keystream
        // (G x the ChaCha work, G x the parallelism -- the grouped loop leaves most CUs idle below ~10^5 packs)
        for (size_t i = gid; i < nvec; i += gsz) {
            W c[TT][P::N];
            const P s = load_s(i);
            rng_draw_pack<F, TT, P::N>(f, ra.rk, ra.r0, ra.r1, (uint64_t)i, (uint64_t)npacks_all, c);
            eval_store(i, s, c);
        }
    } else {
        for (size_t ig = gid; ig < ngroups; ig += gsz) {
            W cg[G][TT][P::N];
            // the group's G packs are fetched (all loads in flight together) BEFORE the keystream is computed, so
            // the ChaCha rounds run in the shadow of the loads instead of each pack waiting for its own after them
            // (split_rng over GF(2^61-1), m=3, t=1: 54.5 vs 59.5 us)
            P sv[G];
            if constexpr (REC) {
                // chain gate: the operand fetch is itself a recombination (k loads and multiply-adds per pack); here
                // the keystream first and the packs one after the other measured faster (105-115 vs 118-121 us for
                // 10^7 elements of GF(2^61-1), m=3, t=1, ChaCha20)
                rng_draw_group<F, TT, P::N>(f, ra.rk, ra.r0, ra.r1, (uint64_t)ig, cg);
#pragma unroll
                for (int u = 0; u < G; ++u) {
                    const size_t i = (size_t)u * ngroups + ig;
                    if (i < nvec) {
                        sv[0] = load_s(i);
                        eval_store(i, sv[0], cg[u]);
                    }
                }
                continue;
            }
#pragma unroll
            for (int u = 0; u < G; ++u) {
                const size_t i = (size_t)u * ngroups + ig;             // stride NG: lanes stay on adjacent packs
                if (i < nvec) sv[u] = load_s(i);
            }
            if constexpr (RNG && T > 0) rng_draw_group<F, TT, P::N>(f, ra.rk, ra.r0, ra.r1, (uint64_t)ig, cg);
#pragma unroll
            for (int u = 0; u < G; ++u) {
                const size_t i = (size_t)u * ngroups + ig;
                if (i < nvec) eval_store(i, sv[u], cg[u]);
            }
        }
    }
    // scalar tail: elements past the last full pack (or everything, if pointers are unaligned)
    constexpr int EPV = P::N * F::EPW;
    const size_t done = nvec * (size_t)EPV;
    for (size_t e = done + gid; e < n; e += gsz) {
        W s;
        if constexpr (REC) {
            const size_t eA = e + yoffA, eB = e + yoffB;
            s = gs.plainA ? ld_elem<F>(gs.rowsA[0], eA) : gate_load_elem<F>(f, gs.rowsA, gs.lamA, gs.kA, eA);
            s = f.mul(s, gs.square ? s
                                   : (gs.plainB ? ld_elem<F>(gs.rowsB[0], eB)
                                                : gate_load_elem<F>(f, gs.rowsB, gs.lamB, gs.kB, eB)));
        } else {
            s = ld_elem<F>(a, e);
            if constexpr (FUSE_MUL) s = f.mul(s, ld_elem<F>(b, e));
        }
        W c[TT];
        if constexpr (RNG && T > 0) {
            W cc[TT][P::N];
            rng_draw_pack<F, T, P::N>(f, ra.rk, ra.r0, ra.r1, (uint64_t)(e / EPV), (uint64_t)((n + EPV - 1) / EPV), cc);
            const int q = (int)((e % EPV) / F::EPW);
#pragma unroll
            for (int j = 0; j < T; ++j) {
                W v = cc[j][0];
#pragma unroll
                for (int qq = 1; qq < P::N; ++qq) v = (qq == q) ? cc[j][qq] : v;
                if constexpr (F::EPW > 1) v = (W)((v >> (8 * (e % F::EPW))) & 0xffu);
                c[j] = v;
            }
        } else {
#pragma unroll
            for (int j = 0; j < T; ++j) c[j] = ld_elem<F>(coef + (size_t)j * cstride, e);
        }
        for (int party = 1; party <= m; ++party) {
            W y = s;
            if constexpr (T > 0) {
                W acc = c[T - 1];
#pragma unroll
                for (int j = T - 2; j >= 0; --j) acc = f.muladd_small(acc, (uint32_t)party, c[j]);
                y = f.muladd_small(acc, (uint32_t)party, s);
            }
            st_elem<F>(out + (size_t)(party - 1) * ostride, e, y);
        }
    }
    if constexpr (RNG) rng_state_release(ra);
}

// materialise the coefficient matrix the fused kernel would draw (tests, debugging, and callers
// that want the coefficients): identical keystream layout.
template <class F, int T>
__global__ __launch_bounds__(BLOCK) void k_rng_coeffs(F f, typename F::elem* __restrict__ coef, size_t cstride,
                                                       size_t nvec, size_t n, RngArgs ra) {
    rng_load_state(ra);
    typedef Pack<typename F::word> P;
    typedef typename MemPack<F>::type MP;
    typedef typename F::word W;
    constexpr int EPV = P::N * F::EPW;
    const size_t gid = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    const size_t gsz = (size_t)gridDim.x * BLOCK;
    const size_t npacks = (n + EPV - 1) / EPV;
    constexpr int G = RngLayout<F, T, P::N>::G;
    const size_t ngroups = (npacks + G - 1) / G;
    for (size_t ig = gid; ig < ngroups; ig += gsz) {
        W cg[G][T][P::N];
        rng_draw_group<F, T, P::N>(f, ra.rk, ra.r0, ra.r1, (uint64_t)ig, cg);
        for (int u = 0; u < G; ++u) {
            const size_t i = (size_t)u * ngroups + ig;
            if (i >= npacks) continue;
            if (i < nvec) {
#pragma unroll
                for (int j = 0; j < T; ++j) {
                    P t_;
#pragma unroll
                    for (int q = 0; q < P::N; ++q) {
                        W v = cg[0][j][q];
#pragma unroll
                        for (int uu = 1; uu < G; ++uu) v = (uu == u) ? cg[uu][j][q] : v;
                        t_.w[q] = v;
                    }
                    stg<true>(reinterpret_cast<MP*>(coef + (size_t)j * cstride) + i, t_);
                }
            } else {
                for (int j = 0; j < T; ++j)
                    for (int q = 0; q < P::N; ++q)
                        for (int b_ = 0; b_ < F::EPW; ++b_) {
                            size_t e = i * EPV + (size_t)q * F::EPW + b_;
                            if (e < n) {
                                W v = cg[0][j][q];
                                for (int uu = 1; uu < G; ++uu) v = (uu == u) ? cg[uu][j][q] : v;
                                if constexpr (F::EPW > 1) v = (W)((v >> (8 * b_)) & 0xffu);
                                st_elem<F>(coef + (size_t)j * cstride, e, v);
                            }
                        }
            }
        }
    }
}

// any degree t: coefficients are re-read per party (they stay in L2/MALL).  With RNG each row j
// is its own keystream (nonce word 1 + j + 1) in the T = 1 layout.
template <class F, bool FUSE_MUL, bool RNG>
__global__ __launch_bounds__(BLOCK) void k_split_any(F f, const typename F::elem* __restrict__ a,
                                                      const typename F::elem* __restrict__ b,
                                                      const typename F::elem* __restrict__ coef, size_t cstride,
                                                      int t, int m, typename F::elem* __restrict__ out,
                                                      size_t ostride, size_t n, RngArgs ra) {
    rng_load_state(ra);
    typedef typename F::word W;
    typedef Pack<W> P;
    constexpr int EPV = P::N * F::EPW;
    const size_t gid = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    const size_t gsz = (size_t)gridDim.x * BLOCK;
    auto coef_at = [&](int j, size_t e) -> W {
        if constexpr (RNG) {
            RngArgs rj = ra;
            rj.rk.nonce[1] += (uint32_t)(j + 1);
            W cc[1][P::N];
            rng_draw_pack<F, 1, P::N>(f, rj.rk, rj.r0, rj.r1, (uint64_t)(e / EPV), (uint64_t)((n + EPV - 1) / EPV), cc);
            const int q = (int)((e % EPV) / F::EPW);
            W v = cc[0][0];
#pragma unroll
            for (int qq = 1; qq < P::N; ++qq) v = (qq == q) ? cc[0][qq] : v;
            if constexpr (F::EPW > 1) v = (W)((v >> (8 * (e % F::EPW))) & 0xffu);
            return v;
        } else {
            return ld_elem<F>(coef + (size_t)j * cstride, e);
        }
    };
    for (size_t e = gid; e < n; e += gsz) {
        W s = ld_elem<F>(a, e);
        if constexpr (FUSE_MUL) s = f.mul(s, ld_elem<F>(b, e));
        for (int party = 1; party <= m; ++party) {
            W acc = coef_at(t - 1, e);
            for (int j = t - 2; j >= 0; --j) acc = f.muladd_small(acc, (uint32_t)party, coef_at(j, e));
            st_elem<F>(out + (size_t)(party - 1) * ostride, e, f.muladd_small(acc, (uint32_t)party, s));
        }
    }
    if constexpr (RNG) rng_state_release(ra);
}

template <class F>
__global__ __launch_bounds__(BLOCK) void k_rng_coeffs_any(F f, typename F::elem* __restrict__ coef, size_t cstride,
                                                           int t, size_t n, RngArgs ra) {
    rng_load_state(ra);
    typedef typename F::word W;
    typedef Pack<W> P;
    constexpr int EPV = P::N * F::EPW;
    const size_t gid = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    const size_t gsz = (size_t)gridDim.x * BLOCK;
    const size_t npacks = (n + EPV - 1) / EPV;
    for (size_t i = gid; i < npacks; i += gsz) {
        for (int j = 0; j < t; ++j) {
            RngArgs rj = ra;
            rj.rk.nonce[1] += (uint32_t)(j + 1);
            W cc[1][P::N];
            rng_draw_pack<F, 1, P::N>(f, rj.rk, rj.r0, rj.r1, (uint64_t)i, (uint64_t)npacks, cc);
            for (int q = 0; q < P::N; ++q)
                for (int b_ = 0; b_ < F::EPW; ++b_) {
                    size_t e = i * EPV + (size_t)q * F::EPW + b_;
                    if (e < n) {
                        W v = cc[0][q];
                        if constexpr (F::EPW > 1) v = (W)((v >> (8 * b_)) & 0xffu);
                        st_elem<F>(coef + (size_t)j * cstride, e, v);
                    }
                }
        }
    }
}

// ---- Lagrange recombination (thresha.py:119-132) ---------------------------
template <class F, int K>
struct RecArgs {
    const typename F::elem* rows[K];
    typename F::word lam[MAXW * K];  // (w, K) prepared constants
};

template <class F, int K, bool NT>
__global__ __launch_bounds__(BLOCK) void k_recombine(F f, RecArgs<F, K> ra, int w,
                                                      typename F::elem* __restrict__ out, size_t ostride,
                                                      size_t nvec, size_t n) {
    typedef Pack<typename F::word> P;
    typedef typename MemPack<F>::type MP;
    const size_t gid = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    const size_t gsz = (size_t)gridDim.x * BLOCK;
    for (size_t i = gid; i < nvec; i += gsz) {
        P x[K];
        {
            // (24-byte elements wave by wave: pays since the dot product runs in 28-bit digits -- the 128-bit arithmetic
            // before that made the kernel VALU-bound over three-limb primes and the LDS round trips a net loss)
            decltype(ldgw_issue<NT>(reinterpret_cast<const MP*>(ra.rows[0]))) rx[K];
#pragma unroll
            for (int j = 0; j < K; ++j) rx[j] = ldgw_issue<NT>(reinterpret_cast<const MP*>(ra.rows[j]) + i);
#pragma unroll
            for (int j = 0; j < K; ++j) x[j] = ldgw_finish(rx[j]);
        }
        for (int r = 0; r < w; ++r) {
            P y;
#pragma unroll
            for (int q = 0; q < P::N; ++q) {
                if constexpr (HasLazyAcc<F>::value) {
                    typename F::lacc s;
                    f.lacc_zero(s);
#pragma unroll
                    for (int j = 0; j < K; ++j) f.lacc_mac(s, ra.lam[r * K + j], x[j].w[q]);
                    y.w[q] = f.lacc_reduce(s);
                } else {
                    typename F::acc s;
                    f.acc_zero(s);
#pragma unroll
                    for (int j = 0; j < K; ++j) f.acc_mac(s, ra.lam[r * K + j], x[j].w[q]);
                    y.w[q] = f.acc_reduce(s);
                }
            }
            stgw<NT>(reinterpret_cast<MP*>(out + (size_t)r * ostride) + i, y);
        }
    }
    const size_t done = nvec * (size_t)(P::N * F::EPW);
    for (size_t e = done + gid; e < n; e += gsz) {
        for (int r = 0; r < w; ++r) {
            typename F::acc s;
            f.acc_zero(s);
#pragma unroll
            for (int j = 0; j < K; ++j) f.acc_mac(s, ra.lam[r * K + j], ld_elem<F>(ra.rows[j], e));
            st_elem<F>(out + (size_t)r * ostride, e, f.acc_reduce(s));
        }
    }
}

template <class F>
struct RecArgsAny {
    const typename F::elem* rows[MAXK_ANY];
    typename F::word lam[MAXK_ANY];  // one output row per launch
};

template <class F>
__global__ __launch_bounds__(BLOCK) void k_recombine_any(F f, RecArgsAny<F> ra, int k,
                                                          typename F::elem* __restrict__ out, size_t n) {
    const size_t gid = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    const size_t gsz = (size_t)gridDim.x * BLOCK;
    for (size_t e = gid; e < n; e += gsz) {
        typename F::acc s;
        f.acc_zero(s);
        for (int j = 0; j < k; ++j) f.acc_mac(s, ra.lam[j], ld_elem<F>(ra.rows[j], e));
        st_elem<F>(out, e, f.acc_reduce(s));
    }
}


// ---- helpers for the exponentiation / inversion kernels ---------------------------------------
// Policies that can carry partially reduced values through a chain of products expose mul_lazy / canon (PM64 for
// p = 2^k - 1, fields.hpp); every other policy multiplies canonically and canon is the identity.
template <class F, class = void>
struct HasLazyMul : std::false_type {};
template <class F>
struct HasLazyMul<F, std::void_t<decltype(std::declval<const F&>().mul_lazy(std::declval<typename F::word>(),
                                                                              std::declval<typename F::word>()))> >
    : std::true_type {};
template <class F>
FF_HD typename F::word ff_mul_lazy(const F& f, typename F::word a, typename F::word b) {
    if constexpr (HasLazyMul<F>::value) return f.mul_lazy(a, b);
    else return f.mul(a, b);
}
template <class F>
FF_HD typename F::word ff_sqr_lazy(const F& f, typename F::word a) {
    if constexpr (HasLazyMul<F>::value) return f.sqr_lazy(a);
    else return f.mul(a, a);
}
template <class F>
FF_HD typename F::word ff_canon(const F& f, typename F::word a) {
    if constexpr (HasLazyMul<F>::value) return f.canon(a);
    else return a;
}
template <class F>
FF_HD typename F::word ff_one(const F&) {
    if constexpr (sizeof(typename F::word) == 24) {
        typename F::word w;
        w.lo = 1;
        w.mid = w.hi = 0;
        return w;
    } else if constexpr (sizeof(typename F::word) == 16) {
        typename F::word w;
        w.lo = 1;
        w.hi = 0;
        return w;
    } else if constexpr (F::EPW == 4) {
        return (typename F::word)0x01010101u;
    } else {
        return (typename F::word)1;
    }
}
template <class F>
FF_HD typename F::word ff_one_elem(const F& f) {       // one in element 0 only (scalar paths of packed fields)
    if constexpr (F::EPW == 4) return (typename F::word)1;
    else return ff_one(f);
}
// zero elements are replaced by one (so that products stay invertible); zm remembers where
template <class F>
FF_HD typename F::word ff_zero_fix(const F&, typename F::word v, uint32_t& zm) {
    if constexpr (sizeof(typename F::word) == 24) {
        zm = (v.lo | v.mid | v.hi) == 0;
        if (zm) v.lo = 1;
        return v;
    } else if constexpr (sizeof(typename F::word) == 16) {
        zm = (v.lo | v.hi) == 0;
        if (zm) v.lo = 1;
        return v;
    } else if constexpr (F::EPW == 4) {
        uint32_t t = ((v & 0x7f7f7f7fu) + 0x7f7f7f7fu) | v;   // bit 7 of each byte set iff byte != 0
        uint32_t z = (~t & 0x80808080u) >> 7;                   // 0x01 where the byte is zero
        zm = z;
        return v | z;
    } else {
        zm = v == 0;
        return zm ? (typename F::word)1 : v;
    }
}
template <class F>
FF_HD typename F::word ff_zero_apply(const F&, typename F::word r, uint32_t zm) {
    if constexpr (sizeof(typename F::word) == 24) {
        if (zm) r.lo = r.mid = r.hi = 0;
        return r;
    } else if constexpr (sizeof(typename F::word) == 16) {
        if (zm) r.lo = r.hi = 0;
        return r;
    } else if constexpr (F::EPW == 4) {
        return r & ~(zm * 0xffu);
    } else {
        return zm ? (typename F::word)0 : r;
    }
}

struct ExpArgs {
    uint64_t e[3];   // public exponent, little-endian limbs (three for the three-limb prime fields)
    int nbits;       // bit length of the exponent (>= 1)
    int post = 0;    // 1: the result is r^3 * a for r = a^e -- ffgpu_pow hands a^(3 e' + 1) over as (e', post) when the chain
    //                  for e' is shorter: the inverse square root exponent (3p - 5) / 4 of p = 3 mod 4 starts "10111...", which
    //                  defeats the leading-run doubling, while e' = (p - 3) / 4 is one long run of ones (finfields.py:1424-1437)
};

// a^e for a public (wave-uniform) exponent e >= 1.  Every branch is on the exponent (scalar); intermediates are
// partially reduced where the policy allows it (ff_mul_lazy), the result is canonical.
//  * A LEADING RUN of r >= 12 set bits -- the inversion exponent q - 2 and the Legendre exponent (q - 1) / 2 of the
//    default primes are almost all ones -- is raised by doubling: a^(2^(2k) - 1) = (a^(2^k - 1))^(2^k) * a^(2^k - 1),
//    r - 1 squarings + about log2(r) + popcount(r) products (2^61 - 3: 60 squarings + 10 products in all, where
//    4-bit windows take 60 + 23 and the binary method 60 + 59).
//  * The remaining bits: plain square-and-multiply when few are set ((p + 1) / 4 = 2^59, the tail of q - 2), otherwise
//    left-to-right SLIDING WINDOWS of up to 4 bits over the odd powers a, a^3, ..., a^15 (one squaring + 7 products
//    to build them; the table lives in registers and is selected by a uniform switch -- no dynamic register
//    indexing, no scratch).
//  * WINDOWS = false leaves the window table out (square-and-multiply for whatever follows the leading run): the
//    batched inverse holds its prefix products across the exponentiation and the eight odd powers cost it 16 registers
//    plus their live ranges; the host picks this form when the exponent's tail is short (ff_pow_lean_ok).
// (the chain itself is generic over an arithmetic `ops` -- value type V, mul, sqr --: the policy's own words with
// ff_mul_lazy / ff_sqr_lazy, or the digit form of fields.hpp DigitChain)
template <bool WINDOWS, class Ops>
__device__ __forceinline__ typename Ops::V ff_pow_chain_core(const Ops& ops, const typename Ops::V a, const ExpArgs& ex);
template <bool WINDOWS, class Ops>
__device__ __forceinline__ typename Ops::V ff_pow_chain(const Ops& ops, const typename Ops::V a, const ExpArgs& ex) {
    typename Ops::V r = ff_pow_chain_core<WINDOWS>(ops, a, ex);
    if (ex.post) r = ops.mul(ops.mul(ops.sqr(r), r), a);        // (wave-uniform)
    return r;
}
template <bool WINDOWS, class Ops>
__device__ __forceinline__ typename Ops::V ff_pow_chain_core(const Ops& ops, const typename Ops::V a, const ExpArgs& ex) {
    typedef typename Ops::V W;
    auto bit = [&](int i) -> uint32_t { return (uint32_t)(ex.e[i >> 6] >> (i & 63)) & 1u; };
    int run = 0;                                 // length of the leading run of set bits (word at a time: scalar clz)
    for (int top = ex.nbits - 1; top >= 0;) {
        const int pos = top & 63;
        const uint64_t inv = ~(ex.e[top >> 6] << (63 - pos));
        int lz = inv ? __builtin_clzll(inv) : 64;
        if (lz > pos + 1) lz = pos + 1;
        run += lz;
        if (lz < pos + 1) break;
        top -= pos + 1;
    }
    W r = a;
    int i = ex.nbits - 2;                        // next bit to consume (the top bit is a itself)
    if (run >= 12) {
        int have = 1;                            // r = a^(2^have - 1)
        for (int b = 30 - __builtin_clz((unsigned)run); b >= 0; --b) {
            W t = r;
            for (int q = 0; q < have; ++q) t = ops.sqr(t);
            r = ops.mul(t, r);
            have *= 2;
            if ((run >> b) & 1) {
                r = ops.mul(ops.sqr(r), a);
                ++have;
            }
        }
        i = ex.nbits - 1 - run;
    }
    int ones = 0;                                // set bits among the remaining bits i..0
    for (int q = 0; q < 3; ++q) {
        const int hi = i - 64 * q;                // highest remaining bit within word q
        if (hi >= 63) ones += __builtin_popcountll(ex.e[q]);
        else if (hi >= 0) ones += __builtin_popcountll(ex.e[q] & ((2ull << hi) - 1));
    }
    if (!WINDOWS || i < 4 || ones <= 8 + (i + 1) / 8) {
        for (; i >= 0; --i) {
            r = ops.sqr(r);
            if (bit(i)) r = ops.mul(r, a);
        }
        return r;
    }
    const W a2 = ops.sqr(a);
    const W t1 = a, t3 = ops.mul(t1, a2), t5 = ops.mul(t3, a2), t7 = ops.mul(t5, a2),
            t9 = ops.mul(t7, a2), t11 = ops.mul(t9, a2), t13 = ops.mul(t11, a2),
            t15 = ops.mul(t13, a2);
    auto odd = [&](uint32_t v) -> W {            // v odd, 1..15, wave-uniform
        switch (v >> 1) {
            case 0: return t1;
            case 1: return t3;
            case 2: return t5;
            case 3: return t7;
            case 4: return t9;
            case 5: return t11;
            case 6: return t13;
            default: return t15;
        }
    };
    while (i >= 0) {
        if (!bit(i)) {
            r = ops.sqr(r);
            --i;
            continue;
        }
        int j = i - 3 > 0 ? i - 3 : 0;
        while (!bit(j)) ++j;                     // the window ends on a set bit: its value is odd
        uint32_t v = 0;
        for (int q = i; q >= j; --q) v = (v << 1) | bit(q);
        for (int q = i; q >= j; --q) r = ops.sqr(r);
        r = ops.mul(r, odd(v));
        i = j - 1;
    }
    return r;
}
template <class F>
struct WordChainOps {                            // the policy's own words, partially reduced where it can (mul_lazy)
    typedef typename F::word V;
    const F& f;
    __device__ __forceinline__ V mul(const V& x, const V& y) const { return ff_mul_lazy(f, x, y); }
    __device__ __forceinline__ V sqr(const V& x) const { return ff_sqr_lazy(f, x); }
};
template <int NL>
struct DigitChainOps {
    typedef typename DigitChain<NL>::val V;
    DigitChain<NL> dc;
    __device__ __forceinline__ V mul(const V& x, const V& y) const { return dc.mul(x, y); }
    __device__ __forceinline__ V sqr(const V& x) const { return dc.sqr(x); }
};
template <class F, bool WINDOWS = true>
__device__ __forceinline__ typename F::word ff_pow(const F& f, typename F::word a, const ExpArgs& ex) {
    const WordChainOps<F> ops{f};
    return ff_canon(f, ff_pow_chain<WINDOWS>(ops, a, ex));
}
// The same in digits where the policy has them (fields.hpp DigitChain: the multi-limb 2^k - c primes) -- ~40 instead of ~100
// instructions per product; used by k_pow (sqrt, inverse sqrt, pow: ~80 products per element, every element).
template <class F, class = void>
struct HasDigitChain : std::false_type {};
template <class F>
struct HasDigitChain<F, std::void_t<decltype(F::CHAIN_MAX_NL)> > : std::true_type {};
template <class F, int NL>
__device__ __forceinline__ bool ff_pow_digits_nl(const F& f, const typename F::word& a, const ExpArgs& ex, typename F::word& out) {
    DigitChainOps<NL> ops;
    if (!f.template chain_setup<NL>(ops.dc)) return false;
    out = f.template chain_out<NL>(ops.dc, ff_pow_chain<true>(ops, f.template chain_in<NL>(ops.dc, a), ex));
    return true;
}
template <class F, int NL>
__device__ __forceinline__ bool ff_pow_digits_from(const F& f, const typename F::word& a, const ExpArgs& ex, typename F::word& out) {
    if constexpr (NL > F::CHAIN_MAX_NL) {
        return false;
    } else {
        if (f.k <= 28u * NL) return ff_pow_digits_nl<F, NL>(f, a, ex, out);     // (f.k is wave-uniform: scalar branches)
        return ff_pow_digits_from<F, NL + 1>(f, a, ex, out);
    }
}
template <class F>
__device__ __forceinline__ typename F::word ff_pow_digits(const F& f, typename F::word a, const ExpArgs& ex) {
    if constexpr (HasDigitChain<F>::value) {
        typename F::word out;
        if (ff_pow_digits_from<F, F::CHAIN_MIN_NL>(f, a, ex, out)) return out;
    }
    return ff_pow(f, a, ex);
}

// ---- out = a^e, public exponent e >= 1 (finfields.py:1159-1187, :1408-1414) ------------------
template <class F, bool NT>
__global__ __launch_bounds__(BLOCK) void k_pow(F f, const typename F::elem* __restrict__ a, ExpArgs ex,
                                                typename F::elem* __restrict__ o, size_t nvec, size_t n) {
    typedef Pack<typename F::word> P;
    typedef typename MemPack<F>::type MP;
    const MP* __restrict__ av = reinterpret_cast<const MP*>(a);
    MP* __restrict__ ov = reinterpret_cast<MP*>(o);
    const size_t gid = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    const size_t gsz = (size_t)gridDim.x * BLOCK;
    for (size_t i = gid; i < nvec; i += gsz) {
        P x = ldg<NT>(av + i);
        P r;
#pragma unroll
        for (int q = 0; q < P::N; ++q) r.w[q] = ff_pow_digits(f, x.w[q], ex);
        stg<NT>(ov + i, r);
    }
    const size_t done = nvec * (size_t)(P::N * F::EPW);
    for (size_t e = done + gid; e < n; e += gsz) st_elem<F>(o, e, ff_pow(f, ld_elem<F>(a, e), ex));
}

// ---- out = a^-1, batched (finfields.py:1278-1281, :1416-1422) ----------------------------------
// Montgomery's trick inside each thread over G independent groups of CH packs: prefix products per group (G chains
// the scheduler interleaves), ONE exponentiation by q-2 of the product of the group totals (sliding windows: 83
// products for a 61-bit prime), the groups' inverses from it, back-substitution per group:
// 3 multiplications per element + (pow + 3 G) / (G CH N).  Only the PREFIX products are kept in registers; the
// operands themselves are read a second time for the back-substitution (they come back from L2 / the Infinity Cache:
// a thread re-reads what it read a few microseconds earlier; HBM traffic stays one read and one write per element),
// which halves the register footprint and lets one exponentiation serve twice as many elements.
// Zero inputs give zero and set *flag (the reference raises ZeroDivisionError; the host wrapper checks the flag): one
// bit per element in a 64-bit mask, and the patch-up of the outputs is skipped by a scalar branch unless some lane of
// the wave met a zero.
//
// (Measured and NOT kept, round 3: pooling the totals of a workgroup -- XOR butterfly over the lanes, wave totals through
// LDS, ONE wave per workgroup raising the pooled total -- replaces 3/4 of the exponentiations by 15 products per thread,
// but the three waves that wait at the barrier leave their SIMDs with one runnable wave: 61.3 us against 56.0 us at
// n = 10^7 over 2^61 - 1.)
template <class F, int CH, int G, bool NT>
__global__ __launch_bounds__(BLOCK) void k_inv_batch(F f, const typename F::elem* __restrict__ a, ExpArgs ex,
                                                      typename F::elem* __restrict__ o, size_t nvec, size_t n,
                                                      int* __restrict__ flag) {
    typedef Pack<typename F::word> P;
    typedef typename MemPack<F>::type MP;
    typedef typename F::word W;
    static_assert(CH * G * P::N <= 64, "zero mask has 64 bits");
    const MP* __restrict__ av = reinterpret_cast<const MP*>(a);
    MP* __restrict__ ov = reinterpret_cast<MP*>(o);
    const size_t gid = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    const size_t gsz = (size_t)gridDim.x * BLOCK;
    uint32_t anyzero = 0;
    for (size_t i0 = gid; i0 < nvec; i0 += gsz * (CH * G)) {
        W pre[G][CH][P::N], tot[G];
        uint64_t zbits = 0;                     // bit ((g * CH + c) * N + q): that operand was zero
#pragma unroll
        for (int g = 0; g < G; ++g) tot[g] = ff_one(f);
#pragma unroll
        for (int c = 0; c < CH; ++c) {
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const size_t j = i0 + (size_t)(g * CH + c) * gsz;
                P t_;
                if (j < nvec) t_ = ldg<false>(av + j);      // (kept cacheable: read again below)
#pragma unroll
                for (int q = 0; q < P::N; ++q) {
                    uint32_t zq;
                    W v = (j < nvec) ? t_.w[q] : ff_one(f);
                    v = ff_zero_fix(f, v, zq);
                    if constexpr (F::EPW > 1) {
                        // packed sub-elements: keep the per-byte mask of this word (at most 4 words per thread here)
                        zbits |= (uint64_t)zq << (8 * (((g * CH + c) * P::N + q) & 7));
                        static_assert(F::EPW == 1 || CH * G * P::N <= 8, "packed fields: 8 words per batch");
                    } else {
                        zbits |= (uint64_t)(zq & 1u) << ((g * CH + c) * P::N + q);
                    }
                    pre[g][c][q] = tot[g];      // product of everything BEFORE this element in its group
                    tot[g] = ff_mul_lazy(f, tot[g], v);
                }
            }
        }
        anyzero |= zbits != 0;
        W all = tot[0];
#pragma unroll
        for (int g = 1; g < G; ++g) all = ff_mul_lazy(f, all, tot[g]);
        const W inv_all = ff_pow(f, all, ex);   // (product of all)^-1
        W ginv[G];
        if constexpr (G == 1) {
            ginv[0] = inv_all;
        } else if constexpr (G == 2) {
            ginv[0] = ff_mul_lazy(f, inv_all, tot[1]);
            ginv[1] = ff_mul_lazy(f, inv_all, tot[0]);
        } else {
            W suf = tot[G - 1], acc = inv_all, sufs[G];
#pragma unroll
            for (int g = G - 2; g >= 0; --g) {
                sufs[g] = suf;                  // product of the totals AFTER g
                suf = ff_mul_lazy(f, suf, tot[g]);
            }
#pragma unroll
            for (int g = 0; g < G; ++g) {
                ginv[g] = g == G - 1 ? acc : ff_mul_lazy(f, acc, sufs[g]);
                acc = ff_mul_lazy(f, acc, tot[g]);       // inv_all * product of the totals up to g
            }
        }
        const bool wave_has_zero = __any(zbits != 0);
#pragma unroll
        for (int c = CH - 1; c >= 0; --c) {
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const size_t j = i0 + (size_t)(g * CH + c) * gsz;
                P t_, r;
                if (j < nvec) t_ = ldg<NT>(av + j);         // second read of the operands (cache hit)
#pragma unroll
                for (int q = P::N - 1; q >= 0; --q) {
                    W v = (j < nvec) ? t_.w[q] : ff_one(f);
                    r.w[q] = ff_canon(f, ff_mul_lazy(f, ginv[g], pre[g][c][q]));
                    if (wave_has_zero) {                     // scalar branch: rare
                        uint32_t zq;
                        v = ff_zero_fix(f, v, zq);
                        r.w[q] = ff_zero_apply(f, r.w[q], zq);
                    }
                    ginv[g] = ff_mul_lazy(f, ginv[g], v);
                }
                if (j < nvec) stg<NT>(ov + j, r);
            }
        }
    }
    const size_t done = nvec * (size_t)(P::N * F::EPW);
    for (size_t e = done + gid; e < n; e += gsz) {
        uint32_t z;
        W v = ff_zero_fix(f, ld_elem<F>(a, e), z);
        if constexpr (F::EPW > 1) z &= 1u;   // a tail element occupies byte 0 only
        anyzero |= z;
        W r = ff_pow(f, v, ex);
        if constexpr (F::EPW > 1) r = z ? (W)0 : r; else r = ff_zero_apply(f, r, z);
        st_elem<F>(o, e, r);
    }
    if (anyzero && flag) atomicOr(flag, 1);
}


// ---- batched inverse over the multi-limb 2^k - c primes, in digits (round 6) ---------------------------------------
// k_inv_batch carries two- and three-limb words: ~100 instructions per product and, at 8 elements per thread, a twelfth of
// an exponentiation of ~100 products per element -- 430 us per 10^7 elements over the 80-bit prime where the one-word primes
// take 45.  Here the whole batch lives in the digit form of fields.hpp DigitChain (NL registers per value, ~47 instructions
// per product): CH elements per thread (32 at NL = 3) share one exponentiation; prefix products, the power and the
// back-substitution never leave the digit domain, elements are converted on the way in and out.  Same structure and zero
// handling as k_inv_batch (second read of the operands from the caches, zeros replaced by one and masked at the end).
// compile-time loop: body(integral_constant<int, I>) for I = FROM, FROM + STEP, ... -- `#pragma unroll` gives up on bodies this
// large (the prefix array then gets a run-time index and moves to scratch memory)
template <int I, int END, int STEP, class Fn>
__device__ __forceinline__ void ff_static_for(Fn&& fn) {
    if constexpr ((STEP > 0 && I < END) || (STEP < 0 && I > END)) {
        fn(std::integral_constant<int, I>());
        ff_static_for<I + STEP, END, STEP>(fn);
    }
}
template <class F, int NL, int CH, bool NT>
__global__ __launch_bounds__(BLOCK) void k_inv_digits(F f, const typename F::elem* __restrict__ a, ExpArgs ex,
                                                       typename F::elem* __restrict__ o, size_t nvec, size_t n,
                                                       int* __restrict__ flag) {
    typedef Pack<typename F::word> P;
    typedef typename MemPack<F>::type MP;
    typedef typename F::word W;
    typedef typename DigitChain<NL>::val V;
    static_assert(P::N == 1 && CH <= 64, "one element per pack; zero mask has 64 bits");
    DigitChainOps<NL> ops;
    const bool ok = f.template chain_setup<NL>(ops.dc);     // (the launcher has checked it)
    const MP* __restrict__ av = reinterpret_cast<const MP*>(a);
    MP* __restrict__ ov = reinterpret_cast<MP*>(o);
    const size_t gid = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    const size_t gsz = (size_t)gridDim.x * BLOCK;
    uint32_t anyzero = 0;
    V one;
#pragma unroll
    for (int d = 0; d < NL; ++d) one.d[d] = d == 0 ? 1u : 0u;
    for (size_t i0 = gid; ok && i0 < nvec; i0 += gsz * CH) {
        V pre[CH], tot = one;
        uint64_t zbits = 0;
        ff_static_for<0, CH, 1>([&](auto ic) {
            constexpr int c = decltype(ic)::value;
            const size_t j = i0 + (size_t)c * gsz;
            W v = ff_one(f);
            if (j < nvec) v = ldg<false>(av + j).w[0];       // (kept cacheable: read again below)
            uint32_t zq;
            v = ff_zero_fix(f, v, zq);
            zbits |= (uint64_t)(zq & 1u) << c;
            pre[c] = tot;                                    // product of everything BEFORE this element
            tot = ops.mul(tot, f.template chain_in<NL>(ops.dc, v));
        });
        anyzero |= zbits != 0;
        V ginv = ff_pow_chain<true>(ops, tot, ex);           // (product of all)^-1
        const bool wave_has_zero = __any(zbits != 0);
        ff_static_for<CH - 1, -1, -1>([&](auto ic) {
            constexpr int c = decltype(ic)::value;
            const size_t j = i0 + (size_t)c * gsz;
            W v = ff_one(f);
            if (j < nvec) v = ldg<NT>(av + j).w[0];          // second read of the operands (cache hit)
            P r;
            r.w[0] = f.template chain_out<NL>(ops.dc, ops.mul(ginv, pre[c]));
            if (wave_has_zero) {                             // scalar branch: rare
                uint32_t zq;
                v = ff_zero_fix(f, v, zq);
                r.w[0] = ff_zero_apply(f, r.w[0], zq);
            }
            if constexpr (c > 0) ginv = ops.mul(ginv, f.template chain_in<NL>(ops.dc, v));
            if (j < nvec) stg<NT>(ov + j, r);
        });
    }
    const size_t done = nvec * (size_t)(P::N * F::EPW);
    for (size_t e = done + gid; e < n; e += gsz) {
        uint32_t z;
        W v = ff_zero_fix(f, ld_elem<F>(a, e), z);
        anyzero |= z;
        st_elem<F>(o, e, ff_zero_apply(f, ff_pow(f, v, ex), z));
    }
    if (anyzero && flag) atomicOr(flag, 1);
}

// ---- batched inverse, one-word fields: full batches without bounds checks ------------------------------------------
// Same arithmetic as k_inv_batch (prefix products per group, one exponentiation per thread, back-substitution with a
// second read of the operands).  What differs is the shape of the code around it:
//   * blocks 0 .. nfull-1 each own BLOCK * CH * G packs and every pack exists: no `j < nvec` predicate anywhere (the
//     predicated loads of k_inv_batch compile to one exec-mask branch per pack and a vmcnt(0) after each second read);
//     the packs past nfull * BLOCK * CH * G (fewer than one block's worth) go to a few extra blocks, one pack per
//     thread, each element raised on its own;
//   * the second reads are issued WIN packs ahead of their use, a scheduling barrier per step keeps the compiler from
//     hoisting them all to the top of the phase (where they sat beside the CH * G prefixes: 10 registers per pack);
//   * the pack addresses of the second pass are formed again from an opaque copy of the stride -- otherwise the
//     2 * CH * G address registers of the first pass stay alive across the exponentiation;
//   * WAVES = the occupancy the register allocator is held to (amdgpu_waves_per_eu);
//   * LEAN: the exponentiation without its window table (ff_pow<F, false>) -- 64 registers less across the one place
//     where all CH * G prefixes are alive; chosen by the host when the exponent's tail is short (every 2^k - c prime).
template <class F, int CH, int G, int WIN, int WAVES, bool LEAN, int WIN1 = 0>
__global__ __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu(WAVES, 8)))
void k_inv_fast(F f, const typename F::elem* __restrict__ a, ExpArgs ex, typename F::elem* __restrict__ o, size_t nvec,
                size_t n, unsigned nfull, int* __restrict__ flag) {
    typedef Pack<typename F::word> P;
    typedef typename MemPack<F>::type MP;
    typedef typename F::word W;
    static_assert(F::EPW == 1 && CH * G * P::N <= 64 && WIN >= 1 && G <= 2, "one-word fields, 64-bit zero mask");
    constexpr int NP = CH * G;
    const MP* __restrict__ av = reinterpret_cast<const MP*>(a);
    MP* __restrict__ ov = reinterpret_cast<MP*>(o);
    uint32_t anyzero = 0;
    auto pack_of = [&](int s_, int& g_, int& c_) { c_ = CH - 1 - s_ / G; g_ = s_ % G; };   // order of the second pass
    if (blockIdx.x < nfull) {
        const size_t gsz = (size_t)nfull * BLOCK;
        const size_t i0 = (size_t)blockIdx.x * BLOCK + threadIdx.x;
        W pre[G][CH][P::N], tot[G];
        uint64_t zbits = 0;
#pragma unroll
        for (int g = 0; g < G; ++g) tot[g] = ff_one(f);
        // first pass in the order c-major, g-minor (the G chains interleave); WIN1 packs are in flight ahead of the
        // products (0 = all CH * G reads issued up front)
        constexpr int W1 = WIN1 > 0 && WIN1 < NP ? WIN1 : NP;
        P ld[W1];
        auto pack1 = [&](int s_) { return (s_ % G) * CH + s_ / G; };          // step -> pack index g * CH + c
#pragma unroll
        for (int s_ = 0; s_ < W1; ++s_) ld[s_] = ldg<false>(av + i0 + (size_t)pack1(s_) * gsz);      // (cacheable: read again below)
#pragma unroll
        for (int s_ = 0; s_ < NP; ++s_) {
            const int g = s_ % G, c = s_ / G;
            const P t_ = ld[s_ % W1];
            if (s_ + W1 < NP) {
                ld[s_ % W1] = ldg<false>(av + i0 + (size_t)pack1(s_ + W1) * gsz);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int q = 0; q < P::N; ++q) {
                uint32_t zq;
                const W v = ff_zero_fix(f, t_.w[q], zq);
                zbits |= (uint64_t)(zq & 1u) << ((g * CH + c) * P::N + q);
                pre[g][c][q] = tot[g];
                tot[g] = ff_mul_lazy(f, tot[g], v);
            }
        }
        anyzero |= zbits != 0;
        W all = tot[0];
        if constexpr (G == 2) all = ff_mul_lazy(f, all, tot[1]);
        const W inv_all = ff_pow<F, !LEAN>(f, all, ex);
        W ginv[G];
        if constexpr (G == 1) {
            ginv[0] = inv_all;
        } else {
            ginv[0] = ff_mul_lazy(f, inv_all, tot[1]);
            ginv[1] = ff_mul_lazy(f, inv_all, tot[0]);
        }
        const bool wave_has_zero = __any(zbits != 0);
        size_t gsz2 = gsz;
        asm volatile("" : "+s"(gsz2));
        P win[WIN];
#pragma unroll
        for (int s_ = 0; s_ < WIN && s_ < NP; ++s_) {
            int g, c;
            pack_of(s_, g, c);
            win[s_ % WIN] = ldg<true>(av + i0 + (size_t)(g * CH + c) * gsz2);
        }
#pragma unroll
        for (int s_ = 0; s_ < NP; ++s_) {
            int g, c;
            pack_of(s_, g, c);
            const P t_ = win[s_ % WIN];
            P r;
            if (s_ + WIN < NP) {
                int g2, c2;
                pack_of(s_ + WIN, g2, c2);
                win[s_ % WIN] = ldg<true>(av + i0 + (size_t)(g2 * CH + c2) * gsz2);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = P::N - 1; q >= 0; --q) {
                W v = t_.w[q];
                r.w[q] = ff_canon(f, ff_mul_lazy(f, ginv[g], pre[g][c][q]));
                if (wave_has_zero) {                     // scalar branch: rare
                    uint32_t zq;
                    v = ff_zero_fix(f, v, zq);
                    r.w[q] = ff_zero_apply(f, r.w[q], zq);
                }
                ginv[g] = ff_mul_lazy(f, ginv[g], v);
            }
            stg<true>(ov + i0 + (size_t)(g * CH + c) * gsz2, r);
        }
    } else {
        // the packs no full block covers (fewer than BLOCK * NP): one pack per thread of the extra blocks, each element
        // raised on its own -- a few microseconds that run beside the full blocks, and no second set of prefix registers
        const size_t j = (size_t)nfull * BLOCK * NP + (size_t)(blockIdx.x - nfull) * BLOCK + threadIdx.x;
        if (j < nvec) {
            const P t_ = ldg<true>(av + j);
            P r;
#pragma unroll
            for (int q = 0; q < P::N; ++q) {
                uint32_t zq;
                const W v = ff_zero_fix(f, t_.w[q], zq);
                anyzero |= zq & 1u;
                r.w[q] = ff_zero_apply(f, ff_pow(f, v, ex), zq);
            }
            stg<true>(ov + j, r);
        }
        const size_t e = nvec * (size_t)P::N + (size_t)(blockIdx.x - nfull) * BLOCK + threadIdx.x;   // past the last whole pack
        if (e < n) {
            uint32_t z;
            const W v = ff_zero_fix(f, ld_elem<F>(a, e), z);
            anyzero |= z & 1u;
            st_elem<F>(o, e, ff_zero_apply(f, ff_pow(f, v, ex), z));
        }
    }
    if (anyzero && flag) atomicOr(flag, 1);
}

// ---- PRSS combination (thresha.py:163-173, 201-217) ---------------------------------------------
// out[h] (+)= sum_{s<ks} sum_{j<d} draw_s[h*d + j] * W[s][j]
// draw_s[i] = the i-th l-byte little-endian chunk of subset s's SHAKE128 output, reduced into
// range(bound) as thresha.PRF.__call__ does (thresha.py:238-266): `% order` (wide reduction,
// l = byte_length + len(key)) or, for a power-of-two bound, a mask.  The XOF itself is sequential
// per key and stays on the host (hashlib); its raw bytes are uploaded once and never boxed.
enum { PRSS_MAXW = 96, PRSS_MAXS = 48 };
template <class F>
struct PrssArgs {
    const uint8_t* streams[PRSS_MAXS];
    typename F::word w[PRSS_MAXW];   // (ks, d) prepared weights f_S(i) * x^(power)
    uint64_t r0, r1;                 // 2^(limb bits) mod p
    int ks, d, l, mask_bits, accumulate;
};

template <class F>
__device__ __forceinline__ typename F::word prss_draw(const F& f, const PrssArgs<F>& pa, const uint8_t* p) {
    typedef typename F::word W;
    constexpr int LB = F::EPW > 1 ? 1 : (int)sizeof(W);   // limb bytes of one element
    const int l = pa.l;
    auto limb = [&](int off, int nbytes) -> W {           // little-endian bytes [off, off+nbytes) as a word
        if constexpr (LB == 24) {
            uint64_t v3[3] = {0, 0, 0};
            for (int b = 0; b < nbytes; ++b) v3[b >> 3] |= (uint64_t)p[off + b] << (8 * (b & 7));
            W w;
            w.lo = v3[0];
            w.mid = v3[1];
            w.hi = v3[2];
            return w;
        } else if constexpr (LB == 16) {
            uint64_t lo = 0, hi = 0;
            for (int b = 0; b < nbytes && b < 8; ++b) lo |= (uint64_t)p[off + b] << (8 * b);
            for (int b = 8; b < nbytes; ++b) hi |= (uint64_t)p[off + b] << (8 * (b - 8));
            W w;
            w.lo = lo;
            w.hi = hi;
            return w;
        } else {
            uint64_t v = 0;
            for (int b = 0; b < nbytes; ++b) v |= (uint64_t)p[off + b] << (8 * b);
            return (W)v;
        }
    };
    if (pa.mask_bits > 0 || l <= LB) {
        // power-of-two bound (or a draw no wider than an element): mask / plain reduction
        W v = limb(0, l < LB ? l : LB);
        if (pa.mask_bits > 0) {
            if constexpr (LB == 24) {
                int mb = pa.mask_bits;
                if (mb < 64) { v.lo &= (1ull << mb) - 1; v.mid = v.hi = 0; }
                else if (mb < 128) { v.mid &= (1ull << (mb - 64)) - 1; v.hi = 0; }
                else if (mb < 192) v.hi &= (1ull << (mb - 128)) - 1;
            } else if constexpr (LB == 16) {
                int mb = pa.mask_bits;
                if (mb < 64) { v.lo &= (1ull << mb) - 1; v.hi = 0; }
                else if (mb < 128) v.hi &= (1ull << (mb - 64)) - 1;
            } else {
                if (pa.mask_bits < 8 * LB) v = (W)((uint64_t)v & ((1ull << pa.mask_bits) - 1));
            }
            return v;            // < bound <= order: canonical
        }
        return f.reduce_raw(v);
    }
    // wide value mod order, limb by limb from the top: r = r * 2^(8 LB) + limb
    W R;
    if constexpr (LB == 24) {
        // 2^192 mod p has up to three limbs for a prime of no special shape: (2^96 mod p)^2, by the policy's own means
        W t96;
        t96.lo = 0;
        t96.mid = 1ull << 32;
        t96.hi = 0;
        t96 = f.reduce_raw(t96);
        R = f.mul(t96, t96);
    } else if constexpr (LB == 16) { R.lo = pa.r0; R.hi = pa.r1; } else { R = (W)pa.r0; }
    int top = (l - 1) / LB * LB;
    W r = f.reduce_raw(limb(top, l - top));
    for (int off = top - LB; off >= 0; off -= LB) r = f.add(f.mul(r, R), f.reduce_raw(limb(off, LB)));
    return r;
}

template <class F>
__global__ __launch_bounds__(BLOCK) void k_prss(F f, PrssArgs<F> pa, typename F::elem* __restrict__ out, size_t n) {
    typedef typename F::word W;
    const size_t gid = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    const size_t gsz = (size_t)gridDim.x * BLOCK;
    for (size_t h = gid; h < n; h += gsz) {
        typename F::acc acc;
        f.acc_zero(acc);
        W total;
        bool have = false;
        int cnt = 0;
        for (int s = 0; s < pa.ks; ++s) {
            const uint8_t* base = pa.streams[s] + (h * (size_t)pa.d) * (size_t)pa.l;
            for (int j = 0; j < pa.d; ++j) {
                W x = prss_draw(f, pa, base + (size_t)j * pa.l);
                f.acc_mac(acc, pa.w[s * pa.d + j], x);
                if (++cnt == 192) {                 // keep the lazy accumulator inside its headroom
                    W part = f.acc_reduce(acc);
                    total = have ? f.add(total, part) : part;
                    have = true;
                    f.acc_zero(acc);
                    cnt = 0;
                }
            }
        }
        W r = f.acc_reduce(acc);
        if (have) r = f.add(total, r);
        if (pa.accumulate) r = f.add(r, ld_elem<F>(out, h));
        st_elem<F>(out, h, r);
    }
}


// ---- PRSS with a COUNTER-MODE PRF (production mode; thresha.py:163-173, 201-217 with PRF := ChaCha) ----------------
// The reference's PRF is one SHAKE128 stream per subset key (thresha.py:238-266): sequential by construction, so in
// parity mode (k_prss above) the host squeezes it and the device idles.  Here every subset key has a ChaCha stream
// (RFC 8439 block function; 256-bit key + 64-bit nonce derived on the host from (PRF key, common input)), addressed by
// block counter, so each lane expands the draws of its own elements.  Sampling rule = the reference's: a draw is l
// little-endian keystream bytes, taken `% bound` (l = byte_length + len(key) for a bound that is not a power of two,
// thresha.py:234-236) or masked for a power-of-two bound -- the value depends on (key, input, index, bound) only, never on
// the field, like the reference's (runtime.py:758-761 evaluates the same PRFs over two fields).
// Public layout (restated in oracle/fforacle.c, oracle/pyoracle.py): LW = ceil(l / 4) keystream words per draw; a TILE
// is TB consecutive blocks holding DPT = min(8, 16 TB / LW) draws, TB in {1,2,3} chosen to waste the least keystream
// (prss_cc_layout).  Draw j of element h (j < d) of a stream: tile = h / DPT, slot = h % DPT, block counters
// (tile * d + j) * TB + b for b < TB, words [slot * LW, slot * LW + LW) of those 16 TB words.
// out[h] (+)= sum_s sum_j W[s][j] * draw_s(h, j).
enum { PRSS_CC_MAXS = 32, PRSS_CC_MAXW = 64, PRSS_CC_MAXDPT = 8, PRSS_CC_MAXTB = 3 };
inline void prss_cc_layout(int l, int* tb, int* dpt) {
    const int lw = (l + 3) / 4;
    int best_tb = 1, best_dpt = 16 / lw < PRSS_CC_MAXDPT ? 16 / lw : PRSS_CC_MAXDPT;
    for (int t = 2; t <= PRSS_CC_MAXTB; ++t) {
        int dp = 16 * t / lw < PRSS_CC_MAXDPT ? 16 * t / lw : PRSS_CC_MAXDPT;
        if (dp * best_tb > best_dpt * t) { best_tb = t; best_dpt = dp; }     // more draws per block: less waste
    }
    *tb = best_tb;
    *dpt = best_dpt;
}
template <class F>
struct PrssCcArgs {
    uint32_t key[PRSS_CC_MAXS][8];
    uint32_t nonce[PRSS_CC_MAXS][2];
    typename F::word w[PRSS_CC_MAXW];   // (ks, d) prepared weights
    uint64_t r0, r1;                    // 2^(limb bits) mod p
    int ks, d, l, mask_bits, accumulate, rounds, tb, dpt;
};

// draw = the l bytes starting at keystream word w0 of this thread's LDS column (word q at col[q * BLOCK]) -> field element
template <class F>
__device__ __forceinline__ typename F::word prss_draw_words(const F& f, const PrssCcArgs<F>& pa, const uint32_t* col, int w0) {
    typedef typename F::word W;
    constexpr int LWD = F::EPW > 1 ? 1 : (int)sizeof(W) / 4;      // keystream words per limb
    const int l = pa.l;
    const int lw = (l + 3) >> 2;
    const uint32_t lastmask = (l & 3) ? ((1u << (8 * (l & 3))) - 1u) : 0xffffffffu;
    auto word32 = [&](int k) -> uint32_t {                          // k-th word of the draw; zero beyond its l bytes
        uint32_t v = k < lw ? col[(size_t)(w0 + k) * BLOCK] : 0u;
        return k == lw - 1 ? (v & lastmask) : v;
    };
    auto word64 = [&](int k) -> uint64_t { return (uint64_t)word32(k) | ((uint64_t)word32(k + 1) << 32); };
    auto limb = [&](int wk) -> W {
        if constexpr (sizeof(W) == 24) {
            W w;
            w.lo = word64(wk);
            w.mid = word64(wk + 2);
            w.hi = word64(wk + 4);
            return w;
        } else if constexpr (sizeof(W) == 16) {
            W w;
            w.lo = word64(wk);
            w.hi = word64(wk + 2);
            return w;
        } else if constexpr (sizeof(W) == 8) {
            return (W)word64(wk);
        } else {
            return (W)word32(wk);
        }
    };
    if (pa.mask_bits > 0 || lw <= LWD) {
        W v = limb(0);
        if (pa.mask_bits > 0) {
            const int mb = pa.mask_bits;
            if constexpr (sizeof(W) == 24) {
                if (mb < 64) { v.lo &= (1ull << mb) - 1; v.mid = v.hi = 0; }
                else if (mb < 128) { v.mid &= (1ull << (mb - 64)) - 1; v.hi = 0; }
                else if (mb < 192) v.hi &= (1ull << (mb - 128)) - 1;
            } else if constexpr (sizeof(W) == 16) {
                if (mb < 64) { v.lo &= (1ull << mb) - 1; v.hi = 0; }
                else if (mb < 128) v.hi &= (1ull << (mb - 64)) - 1;
            } else {
                if (mb < 8 * (int)sizeof(W)) v = (W)((uint64_t)v & ((1ull << mb) - 1));
            }
            return v;            // < bound <= order: canonical
        }
        return f.reduce_raw(v);
    }
    W R;
    if constexpr (sizeof(W) == 24) {
        W t96;
        t96.lo = 0;
        t96.mid = 1ull << 32;
        t96.hi = 0;
        t96 = f.reduce_raw(t96);
        R = f.mul(t96, t96);
    } else if constexpr (sizeof(W) == 16) { R.lo = pa.r0; R.hi = pa.r1; } else { R = (W)pa.r0; }
    const int topw = (lw - 1) / LWD * LWD;
    W r = f.reduce_raw(limb(topw));
    for (int wk = topw - LWD; wk >= 0; wk -= LWD) r = f.add(f.mul(r, R), f.reduce_raw(limb(wk)));
    return r;
}

template <class F>
__global__ __launch_bounds__(BLOCK) void k_prss_chacha(F f, PrssCcArgs<F> pa, typename F::elem* __restrict__ out, size_t n) {
    typedef typename F::word W;
    // the keystream of a tile goes through LDS because LW is a run-time value: a draw's words sit at run-time (wave-uniform)
    // offsets; one column per thread (word q of thread t at [q * BLOCK + t]: conflict-free), no barrier anywhere
    __shared__ uint32_t ksm[PRSS_CC_MAXTB * 16 * BLOCK];
    uint32_t* col = ksm + threadIdx.x;
    const size_t tile = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    const size_t h0 = tile * (size_t)pa.dpt;
    if (h0 >= n) return;
    const int lw = (pa.l + 3) >> 2;
    typename F::acc acc[PRSS_CC_MAXDPT];
#pragma unroll
    for (int i = 0; i < PRSS_CC_MAXDPT; ++i) f.acc_zero(acc[i]);
    for (int s = 0; s < pa.ks; ++s) {
        for (int j = 0; j < pa.d; ++j) {
            const uint64_t c0 = ((uint64_t)tile * (uint64_t)pa.d + (uint64_t)j) * (uint64_t)pa.tb;
            for (int b = 0; b < pa.tb; ++b) {
                uint32_t blk[16];
                const uint64_t ctr = c0 + (uint64_t)b;
                chacha_block(pa.key[s], (uint32_t)ctr, (uint32_t)(ctr >> 32), pa.nonce[s][0], pa.nonce[s][1], pa.rounds, blk);
#pragma unroll
                for (int q = 0; q < 16; ++q) col[(size_t)(b * 16 + q) * BLOCK] = blk[q];
            }
            const W wt = pa.w[s * pa.d + j];
#pragma unroll
            for (int i = 0; i < PRSS_CC_MAXDPT; ++i)
                if (i < pa.dpt) f.acc_mac(acc[i], wt, prss_draw_words(f, pa, col, i * lw));
        }
    }
#pragma unroll
    for (int i = 0; i < PRSS_CC_MAXDPT; ++i)
        if (i < pa.dpt && h0 + (size_t)i < n) {
            W r = f.acc_reduce(acc[i]);
            if (pa.accumulate) r = f.add(r, ld_elem<F>(out, h0 + (size_t)i));
            st_elem<F>(out, h0 + (size_t)i, r);
        }
}


// ---- reductions: out[0] = sum_i a[i] * b[i]   (b == nullptr: sum_i a[i]) ---------------------------
// The local part of an inner product of secret-shared vectors (runtime.in_prod: sum(map(mul, x, y))
// then ONE reshare) and of FieldArray.sum().  Two launches: every workgroup reduces a slice (lazy
// multiply-accumulate per thread, flushed every 192 terms, then an LDS tree with field additions) into
// partial[blockIdx]; a single workgroup then folds the partials.
enum { DOT_MAX_BLOCKS = 1024 };

// cross-lane exchange of a field word (4, 8 or 16 bytes) within a wave, 32 bits at a time
template <class W>
__device__ __forceinline__ W wave_shfl_xor(const W& v, int mask) {
    static_assert(sizeof(W) % 4 == 0, "word size");
    union {
        W w;
        int d[sizeof(W) / 4];
    } in, outv;
    in.w = v;
#pragma unroll
    for (int q = 0; q < (int)(sizeof(W) / 4); ++q) outv.d[q] = __shfl_xor(in.d[q], mask, 64);
    return outv.w;
}

template <class F>
__device__ __forceinline__ typename F::word block_reduce_add(const F& f, typename F::word v, typename F::word* sm) {
    sm[threadIdx.x] = v;
    __syncthreads();
    for (int s = BLOCK / 2; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) sm[threadIdx.x] = f.add(sm[threadIdx.x], sm[threadIdx.x + s]);
        __syncthreads();
    }
    return sm[0];
}

template <class F, bool HAS_B>
__global__ __launch_bounds__(BLOCK) void k_dot_partial(F f, const typename F::elem* __restrict__ a,
                                                        const typename F::elem* __restrict__ b,
                                                        typename F::word* __restrict__ partial, size_t nvec, size_t n) {
    typedef Pack<typename F::word> P;
    typedef typename MemPack<F>::type MP;
    typedef typename F::word W;
    __shared__ W sm[BLOCK];
    const MP* __restrict__ av = reinterpret_cast<const MP*>(a);
    const MP* __restrict__ bv = reinterpret_cast<const MP*>(b);
    const size_t gid = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    const size_t gsz = (size_t)gridDim.x * BLOCK;
    const W one = f.prep(ff_one(f));
    typename F::acc acc;
    f.acc_zero(acc);
    W total = W();
    bool have = false;
    int cnt = 0;
    auto flush = [&]() {
        W part = f.acc_reduce(acc);
        total = have ? f.add(total, part) : part;
        have = true;
        f.acc_zero(acc);
        cnt = 0;
    };
    for (size_t i = gid; i < nvec; i += gsz) {
        P x = ldg<true>(av + i);
        P y;
        if constexpr (HAS_B) y = ldg<true>(bv + i);
#pragma unroll
        for (int q = 0; q < P::N; ++q) {
            if constexpr (F::EPW > 1) {
                // packed GF(2^n<=8): four independent byte lanes per word, xor-accumulated
                if constexpr (HAS_B) acc.a ^= f.mul(x.w[q], y.w[q]); else acc.a ^= x.w[q];
            } else {
                if constexpr (HAS_B) f.acc_mac(acc, f.prep(x.w[q]), y.w[q]);
                else f.acc_mac(acc, one, x.w[q]);
            }
        }
        cnt += P::N;
        if (cnt >= 192) flush();
    }
    const size_t done = nvec * (size_t)(P::N * F::EPW);
    for (size_t e = done + gid; e < n; e += gsz) {
        if constexpr (F::EPW > 1) {
            if constexpr (HAS_B) acc.a ^= f.mul(ld_elem<F>(a, e), ld_elem<F>(b, e)); else acc.a ^= ld_elem<F>(a, e);
        } else {
            if constexpr (HAS_B) f.acc_mac(acc, f.prep(ld_elem<F>(a, e)), ld_elem<F>(b, e));
            else f.acc_mac(acc, one, ld_elem<F>(a, e));
        }
        if (++cnt >= 192) flush();
    }
    W r = f.acc_reduce(acc);
    if (have) r = f.add(total, r);
    r = block_reduce_add(f, r, sm);
    if (threadIdx.x == 0) partial[blockIdx.x] = r;
}

template <class F>
__global__ __launch_bounds__(BLOCK) void k_dot_final(F f, const typename F::word* __restrict__ partial, int nparts,
                                                      typename F::elem* __restrict__ out) {
    typedef typename F::word W;
    __shared__ W sm[BLOCK];
    W r = W();
    bool have = false;
    for (int i = threadIdx.x; i < nparts; i += BLOCK) {
        r = have ? f.add(r, partial[i]) : partial[i];
        have = true;
    }
    r = block_reduce_add(f, r, sm);     // threads without a partial contribute the zero word
    if (threadIdx.x == 0) {
        if constexpr (F::EPW > 1) {
            // packed GF(2^n<=8): fold the four byte lanes of the word into one element
            W w = r;
            w = (w ^ (w >> 8) ^ (w >> 16) ^ (w >> 24)) & 0xffu;
            out[0] = (typename F::elem)w;
        } else {
            st_elem<F>(out, 0, r);
        }
    }
}




// ---- square roots for p = 1 mod 4: Cipolla-Lehmer, exactly as finfields.py:447-470 -----------------
// Per element: smallest b >= 1 with b^2 - 4a a non-residue (Legendre symbol by exponentiation; lanes
// that found theirs idle until the wave is done: 2 candidates expected), then X^((p+1)/2) mod
// X^2 - bX + a by the reference's ladder (public exponent: uniform control flow).  a = 0 -> 0.
// Non-residues give the same (meaningless) value as the reference, which does not test either.
template <class F>
FF_HD typename F::word ff_small(const F&, uint32_t b) {
    if constexpr (sizeof(typename F::word) == 24) {
        typename F::word w;
        w.lo = b;
        w.mid = w.hi = 0;
        return w;
    } else if constexpr (sizeof(typename F::word) == 16) {
        typename F::word w;
        w.lo = b;
        w.hi = 0;
        return w;
    } else {
        return (typename F::word)b;
    }
}
template <class F>
FF_HD bool ff_is_zero(const F& f, typename F::word v) {
    uint32_t zm;
    ff_zero_fix(f, v, zm);
    return zm & 1;
}

template <class F>
__global__ __launch_bounds__(BLOCK) void k_sqrt_cl(F f, const typename F::elem* __restrict__ a, ExpArgs eleg, ExpArgs elad,
                                                    typename F::elem* __restrict__ o, size_t n) {
    typedef typename F::word W;
    const size_t gid = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    const size_t gsz = (size_t)gridDim.x * BLOCK;
    const W one = ff_one_elem(f);
    const W m1 = f.neg(one);
    for (size_t e = gid; e < n; e += gsz) {
        const W x = ld_elem<F>(a, e);
        if (ff_is_zero(f, x)) {
            st_elem<F>(o, e, x);
            continue;
        }
        const W x2 = f.add(x, x);
        const W x4 = f.add(x2, x2);
        uint32_t b = 1;
        W bw;
        for (;;) {
            bw = ff_small(f, b);
            W d = f.sub(f.mul(bw, bw), x4);
            if (!ff_is_zero(f, d) && ff_is_zero(f, f.sub(ff_pow(f, d, eleg), m1))) break;
            ++b;
        }
        W u = f.sub(one, one), v = one;
        for (int i = elad.nbits - 1; i >= 0; --i) {
            W u2 = f.mul(u, u);
            W nu = f.add(f.mul(f.add(u, u), v), f.mul(bw, u2));
            v = f.sub(f.mul(v, v), f.mul(x, u2));
            u = nu;
            if ((elad.e[i >> 6] >> (i & 63)) & 1) {
                W t = f.add(v, f.mul(bw, u));
                v = f.neg(f.mul(x, u));
                u = t;
            }
        }
        st_elem<F>(o, e, v);
    }
}

// ---- Gaussian elimination on a batch of (n x ncols) matrices, in place -----------------------------
// finfields.py:872-955 (gauss_solve / gauss_inv / gauss_det behind np.linalg.solve / inv / det).
// Two launches per pivot column k, for every matrix of the batch at once:
//   k_gauss_pivot (one workgroup per matrix): first row x >= k with A[x][k] != 0 -- the reference's pivot
//     rule, which fixes the swap count and hence its (unsigned) determinant --, swap rows
//     k and x, scale row k by 1/pivot, det *= pivot; no such row: singular flag (det = 0), matrix skipped;
//   k_gauss_elim (2-D grid): A[i][j] -= A[i][k] * A[k][j] for j > k and i != k (solve: Gauss-Jordan, the
//     solution ends up in columns n..ncols-1) or i > k (det).
// The result of solve/inv is unique, so the elimination order need not follow the reference's LU + back
// substitution; the determinant follows its pivot rule exactly.
template <class F>
__global__ __launch_bounds__(BLOCK) void k_gauss_pivot(F f, typename F::elem* __restrict__ A, int n, int ncols, int k,
                                                        ExpArgs ex, typename F::elem* __restrict__ det,
                                                        int* __restrict__ sing) {
    typedef typename F::word W;
    const size_t b = blockIdx.x;
    if (sing[b]) return;
    typename F::elem* M = A + b * (size_t)n * ncols;
    __shared__ int piv;
    if (threadIdx.x == 0) piv = n;
    __syncthreads();
    for (int i = k + threadIdx.x; i < n; i += BLOCK) {
        uint32_t zm;
        ff_zero_fix(f, ld_elem<F>(M, (size_t)i * ncols + k), zm);
        if (!(zm & 1)) {
            atomicMin(&piv, i);
            break;                                   // later rows of this thread are larger
        }
    }
    __syncthreads();
    const int x = piv;
    if (x == n) {
        if (threadIdx.x == 0) {
            sing[b] = 1;
            if (det) st_elem<F>(det, b, ld_elem<F>(M, (size_t)k * ncols + k));   // = 0
        }
        return;
    }
    const W pv = ld_elem<F>(M, (size_t)x * ncols + k);
    const W inv = ff_pow(f, pv, ex);
    __syncthreads();                                 // every thread has read the pivot before row x changes
    for (int j = k + threadIdx.x; j < ncols; j += BLOCK) {
        W a = ld_elem<F>(M, (size_t)k * ncols + j);
        W c = ld_elem<F>(M, (size_t)x * ncols + j);
        if (x != k) st_elem<F>(M, (size_t)x * ncols + j, a);
        st_elem<F>(M, (size_t)k * ncols + j, j == k ? ff_one_elem(f) : f.mul(c, inv));
    }
    if (det && threadIdx.x == 0) st_elem<F>(det, b, k == 0 ? pv : f.mul(ld_elem<F>(det, b), pv));
}

template <class F, int TI>
__global__ __launch_bounds__(BLOCK) void k_gauss_elim(F f, typename F::elem* __restrict__ A, int n, int ncols, int k,
                                                       int lower_only, const int* __restrict__ sing) {
    typedef typename F::word W;
    const size_t b = blockIdx.z;
    if (sing[b]) return;
    typename F::elem* M = A + b * (size_t)n * ncols;
    const int j = k + 1 + blockIdx.x * BLOCK + threadIdx.x;
    if (j >= ncols) return;
    const int i0 = (lower_only ? k + 1 : 0) + blockIdx.y * TI;
    const W r = ld_elem<F>(M, (size_t)k * ncols + j);
#pragma unroll
    for (int q = 0; q < TI; ++q) {
        const int i = i0 + q;
        if (i < n && i != k) {
            W m = ld_elem<F>(M, (size_t)i * ncols + k);          // wave-uniform address: one broadcast load
            W a = ld_elem<F>(M, (size_t)i * ncols + j);
            st_elem<F>(M, (size_t)i * ncols + j, f.sub(a, f.mul(m, r)));
        }
    }
}

// ---- small public matrix applied to every group of g consecutive elements --------------------------
// out[i*r + a] = bias[a] + sum_{c<g} M[a][c] * in[i*g + c],   a < r,  i < ngroups   (r, g <= 16)
// The array-of-structs sibling of k_recombine: finfields `A @ x[..., np.newaxis]` with a public A
// (demos/np_aes.py:40: the 8x8 GF(2) matrix of the S-box applied to the 8 bit-shares of every byte),
// and runtime.np_from_bits (runtime.py:4475-4484: sum_j x_j * 2^j over the last axis, r = 1).
enum { GM_MAX = 16 };
template <class F>
struct GroupMatArgs {
    typename F::word m[GM_MAX * GM_MAX];   // prepared, row-major (r, g)
    typename F::word bias[GM_MAX];
    int r, g;
};

template <class F>
__global__ __launch_bounds__(BLOCK) void k_group_matvec(F f, GroupMatArgs<F> ga, const typename F::elem* __restrict__ in,
                                                         typename F::elem* __restrict__ out, size_t ngroups) {
    typedef typename F::word W;
    const size_t gid = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    const size_t gsz = (size_t)gridDim.x * BLOCK;
    for (size_t i = gid; i < ngroups; i += gsz) {
        W x[GM_MAX];
#pragma unroll
        for (int c = 0; c < GM_MAX; ++c)
            if (c < ga.g) x[c] = ld_elem<F>(in, i * (size_t)ga.g + c);
        for (int a = 0; a < ga.r; ++a) {
            typename F::acc s;
            f.acc_zero(s);
#pragma unroll
            for (int c = 0; c < GM_MAX; ++c)
                if (c < ga.g) f.acc_mac(s, ga.m[a * ga.g + c], x[c]);
            st_elem<F>(out, i * (size_t)ga.r + a, f.add(f.acc_reduce(s), ga.bias[a]));
        }
    }
}


// GF(2^n <= 8), groups of 8 bytes (the bit shares of one byte, np_aes / np_from_bits): one 8-byte load per
// lane, R = 8 -> one 8-byte store, R = 1 -> one byte.  Same arithmetic as k_group_matvec.
template <class F, int R>
__global__ __launch_bounds__(BLOCK) void k_group8_bytes(F f, GroupMatArgs<F> ga, const uint8_t* __restrict__ in,
                                                         uint8_t* __restrict__ out, size_t ngroups) {
    typedef typename F::word W;
    const size_t gid = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    const size_t gsz = (size_t)gridDim.x * BLOCK;
    for (size_t i = gid; i < ngroups; i += gsz) {
        const uint64_t v = __builtin_nontemporal_load(reinterpret_cast<const uint64_t*>(in) + i);
        W x[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) x[c] = (W)((v >> (8 * c)) & 0xffu);
        uint64_t o = 0;
#pragma unroll
        for (int a = 0; a < R; ++a) {
            typename F::acc s;
            f.acc_zero(s);
#pragma unroll
            for (int c = 0; c < 8; ++c) f.acc_mac(s, ga.m[a * 8 + c], x[c]);
            o |= (uint64_t)(f.add(f.acc_reduce(s), ga.bias[a]) & 0xffu) << (8 * a);
        }
        if constexpr (R == 8) __builtin_nontemporal_store(o, reinterpret_cast<uint64_t*>(out) + i);
        else out[i] = (uint8_t)o;
    }
}

}  // namespace ffgpu

#include "matmul.hpp"   // dense, skinny and matrix-core products
#include "launch.hpp"   // host side: FieldOps table + launchers
