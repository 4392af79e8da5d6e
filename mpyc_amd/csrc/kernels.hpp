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
// (profiles/r01_tuning.md).  FFGPU_NT=0 turns the hint off.
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
__global__ __launch_bounds__(BLOCK) void k_ew2(F f, const typename F::elem* __restrict__ a,
                                                const typename F::elem* __restrict__ b,
                                                typename F::elem* __restrict__ o, size_t nvec, size_t n) {
    typedef Pack<typename F::word> P;
    typedef typename MemPack<F>::type MP;
    const MP* __restrict__ av = reinterpret_cast<const MP*>(a);
    const MP* __restrict__ bv = reinterpret_cast<const MP*>(b);
    MP* __restrict__ ov = reinterpret_cast<MP*>(o);
    const size_t gid = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    const size_t gsz = (size_t)gridDim.x * BLOCK;
    for (size_t i = gid; i < nvec; i += gsz) {
        P x = ldg<NT>(av + i);
        P y = ldg<NT>(bv + i);
        P r;
#pragma unroll
        for (int q = 0; q < P::N; ++q) r.w[q] = ew_apply<F, OP>(f, x.w[q], y.w[q]);
        stg<NT>(ov + i, r);
    }
    // scalar tail (n not a multiple of the pack size, or unaligned pointers: nvec == 0)
    const size_t done = nvec * (size_t)(P::N * F::EPW);
    for (size_t e = done + gid; e < n; e += gsz) {
        st_elem<F>(o, e, ew_apply<F, OP>(f, ld_elem<F>(a, e), ld_elem<F>(b, e)));
    }
}

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
        P x = ldg<NT>(av + i);
        P r;
#pragma unroll
        for (int q = 0; q < P::N; ++q) r.w[q] = ew_apply<F, OP>(f, x.w[q], s);
        stg<NT>(ov + i, r);
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
        P x = ldg<NT>(av + i);
        P y = ldg<NT>(bv + i);
        P z = ldg<NT>(cv + i);
        P r;
#pragma unroll
        for (int q = 0; q < P::N; ++q) r.w[q] = f.muladd(x.w[q], y.w[q], z.w[q]);
        stg<NT>(ov + i, r);
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
        P pz = ldg<NT>(reinterpret_cast<const MP*>(z) + i), px = ldg<NT>(reinterpret_cast<const MP*>(x) + i);
        P py = ldg<NT>(reinterpret_cast<const MP*>(y) + i), pd = ldg<NT>(reinterpret_cast<const MP*>(d) + i);
        P pe = ldg<NT>(reinterpret_cast<const MP*>(e) + i), r;
#pragma unroll
        for (int q = 0; q < P::N; ++q) r.w[q] = comb(pz.w[q], px.w[q], py.w[q], pd.w[q], pe.w[q]);
        stg<NT>(reinterpret_cast<MP*>(o) + i, r);
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

template <class F, bool NT>
__device__ __forceinline__ Pack<typename F::word> gate_load(const F& f, const typename F::elem* const* rows,
                                                            const typename F::word* lam, int k, size_t i) {
    typedef Pack<typename F::word> P;
    typedef typename MemPack<F>::type MP;
    typename F::acc acc[P::N];
#pragma unroll
    for (int q = 0; q < P::N; ++q) f.acc_zero(acc[q]);
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
                for (int q = 0; q < P::N; ++q) f.acc_mac(acc[q], lam[j0 + u], x[u].w[q]);
            }
    }
    P r;
#pragma unroll
    for (int q = 0; q < P::N; ++q) r.w[q] = f.acc_reduce(acc[q]);
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
            s = ldg<NT>(av + i);
            if constexpr (FUSE_MUL) s2 = ldg<NT>(bv + i);
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
#pragma unroll
            for (int j = 0; j < T; ++j) {
                P t_ = ldg<NT>(reinterpret_cast<const MP*>(coef + (size_t)j * cstride) + i);
#pragma unroll
                for (int q = 0; q < P::N; ++q) c[j][q] = t_.w[q];
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
                stg<NT>(reinterpret_cast<MP*>(out + (size_t)(party - 1) * ostride) + i, y);
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
                stg<NT>(reinterpret_cast<MP*>(out + (size_t)(party - 1) * ostride) + i, y);
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
#pragma unroll
        for (int j = 0; j < K; ++j) x[j] = ldg<NT>(reinterpret_cast<const MP*>(ra.rows[j]) + i);
        for (int r = 0; r < w; ++r) {
            P y;
#pragma unroll
            for (int q = 0; q < P::N; ++q) {
                typename F::acc s;
                f.acc_zero(s);
#pragma unroll
                for (int j = 0; j < K; ++j) f.acc_mac(s, ra.lam[r * K + j], x[j].w[q]);
                y.w[q] = f.acc_reduce(s);
            }
            stg<NT>(reinterpret_cast<MP*>(out + (size_t)r * ostride) + i, y);
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
template <class F, bool WINDOWS = true>
__device__ __forceinline__ typename F::word ff_pow(const F& f, typename F::word a, const ExpArgs& ex) {
    typedef typename F::word W;
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
            for (int q = 0; q < have; ++q) t = ff_sqr_lazy(f, t);
            r = ff_mul_lazy(f, t, r);
            have *= 2;
            if ((run >> b) & 1) {
                r = ff_mul_lazy(f, ff_sqr_lazy(f, r), a);
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
            r = ff_sqr_lazy(f, r);
            if (bit(i)) r = ff_mul_lazy(f, r, a);
        }
        return ff_canon(f, r);
    }
    const W a2 = ff_sqr_lazy(f, a);
    const W t1 = a, t3 = ff_mul_lazy(f, t1, a2), t5 = ff_mul_lazy(f, t3, a2), t7 = ff_mul_lazy(f, t5, a2),
            t9 = ff_mul_lazy(f, t7, a2), t11 = ff_mul_lazy(f, t9, a2), t13 = ff_mul_lazy(f, t11, a2),
            t15 = ff_mul_lazy(f, t13, a2);
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
            r = ff_sqr_lazy(f, r);
            --i;
            continue;
        }
        int j = i - 3 > 0 ? i - 3 : 0;
        while (!bit(j)) ++j;                     // the window ends on a set bit: its value is odd
        uint32_t v = 0;
        for (int q = i; q >= j; --q) v = (v << 1) | bit(q);
        for (int q = i; q >= j; --q) r = ff_sqr_lazy(f, r);
        r = ff_mul_lazy(f, r, odd(v));
        i = j - 1;
    }
    return ff_canon(f, r);
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
        for (int q = 0; q < P::N; ++q) r.w[q] = ff_pow(f, x.w[q], ex);
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


// ---- dense matrix product C = A @ B over the field (finfields.py:1126-1135, runtime.py:2531) -----
// Classic LDS-tiled product, but the inner operation is the field's lazily reduced multiply-
// accumulate (acc_mac: 128/256-bit products summed unreduced, one reduction per FLUSH products), so
// the cost per MAC is the 4 (16) v_mad_u64_u32 of the product plus carry adds.  Integer-ALU bound.
// Workgroup 16x16 threads, tile 64 x 32 (one-limb fields: 4x2 per thread) or 32 x 32 (two-limb:
// 2x2 per thread), K step 16 staged through LDS; A is stored transposed in LDS so that both operand
// reads are row-contiguous.  Ragged edges are zero-filled on load and masked on store.
template <class W>
__device__ __forceinline__ W ff_keep_if(W v, bool ok) {
    if constexpr (sizeof(W) == 24) {
        v.lo = ok ? v.lo : 0;
        v.mid = ok ? v.mid : 0;
        v.hi = ok ? v.hi : 0;
        return v;
    } else if constexpr (sizeof(W) == 16) {
        v.lo = ok ? v.lo : 0;
        v.hi = ok ? v.hi : 0;
        return v;
    } else {
        return ok ? v : (W)0;
    }
}

// kchunk > 0: split-K -- slice blockIdx.z multiplies columns [z*kchunk, (z+1)*kchunk) of A by the matching rows of
// B into its own (M x N) slab of C (slab stride zstride elements); k_splitk_sum adds the slabs.  Shapes whose
// output gives fewer tiles than the chip has CUs (a batch of 64 activations times a 4096^2 weight matrix) would
// otherwise leave most of it idle.
template <class F, int TM, int TN>
__global__ __launch_bounds__(BLOCK) void k_matmul(F f, const typename F::elem* __restrict__ A, size_t lda,
                                                   const typename F::elem* __restrict__ B, size_t ldb,
                                                   typename F::elem* __restrict__ C, size_t ldc, int M, int K, int N,
                                                   int kchunk, size_t zstride) {
    typedef typename F::word W;
    static_assert(F::EPW == 1, "packed fields use the byte-wise instantiation");
    if (kchunk > 0) {
        const int kz = blockIdx.z * kchunk;
        A += kz;
        B += (size_t)kz * ldb;
        C += (size_t)blockIdx.z * zstride;
        K = K - kz < kchunk ? K - kz : kchunk;
    }
    constexpr int BK = 16, BM = 16 * TM, BN = 16 * TN, FLUSH = 192;
    __shared__ W As[BK][BM + 1];
    __shared__ W Bs[BK][BN + 1];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    typename F::acc acc[TM][TN];
    W tot[TM][TN];
    bool have = false;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) f.acc_zero(acc[i][j]);
    int since = 0;
    for (int k0 = 0; k0 < K; k0 += BK) {
        // stage A (BM x BK) transposed and B (BK x BN)
        for (int idx = threadIdx.x; idx < BM * BK; idx += BLOCK) {
            int mm = idx / BK, kk = idx % BK;
            int gm = m0 + mm, gk = k0 + kk;
            const bool ok = gm < M && gk < K;      // out-of-range: read element 0, then zero it
            As[kk][mm] = ff_keep_if<W>(f.prep(ld_elem<F>(A, ok ? (size_t)gm * lda + gk : 0)), ok);
        }
        for (int idx = threadIdx.x; idx < BK * BN; idx += BLOCK) {
            int kk = idx / BN, nn = idx % BN;
            int gk = k0 + kk, gn = n0 + nn;
            const bool ok = gk < K && gn < N;
            Bs[kk][nn] = ff_keep_if<W>(ld_elem<F>(B, ok ? (size_t)gk * ldb + gn : 0), ok);
        }
        __syncthreads();
#pragma unroll 4
        for (int kk = 0; kk < BK; ++kk) {
            W a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = As[kk][ty + 16 * i];
#pragma unroll
            for (int j = 0; j < TN; ++j) b[j] = Bs[kk][tx + 16 * j];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) f.acc_mac(acc[i][j], a[i], b[j]);
        }
        __syncthreads();
        since += BK;
        if (since >= FLUSH) {   // keep the unreduced accumulators inside their headroom (2^8 products)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    W part = f.acc_reduce(acc[i][j]);
                    tot[i][j] = have ? f.add(tot[i][j], part) : part;
                    f.acc_zero(acc[i][j]);
                }
            have = true;
            since = 0;
        }
    }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            int gm = m0 + ty + 16 * i, gn = n0 + tx + 16 * j;
            if (gm < M && gn < N) {
                W r = f.acc_reduce(acc[i][j]);
                if (have) r = f.add(tot[i][j], r);
                st_elem<F>(C, (size_t)gm * ldc + gn, r);
            }
        }
}

// GF(2^n <= 8): one element per byte, computed element-wise (word = one element in the low byte)
template <class F>
__global__ __launch_bounds__(BLOCK) void k_matmul_bytes(F f, const uint8_t* __restrict__ A, size_t lda,
                                                         const uint8_t* __restrict__ B, size_t ldb,
                                                         uint8_t* __restrict__ C, size_t ldc, int M, int K, int N) {
    constexpr int BK = 16, BM = 32, BN = 32;
    __shared__ uint8_t As[BK][BM + 4];
    __shared__ uint8_t Bs[BK][BN + 4];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    uint32_t acc[2][2] = {{0, 0}, {0, 0}};
    for (int k0 = 0; k0 < K; k0 += BK) {
        for (int idx = threadIdx.x; idx < BM * BK; idx += BLOCK) {
            int mm = idx / BK, kk = idx % BK;
            int gm = m0 + mm, gk = k0 + kk;
            As[kk][mm] = (gm < M && gk < K) ? A[(size_t)gm * lda + gk] : 0;
        }
        for (int idx = threadIdx.x; idx < BK * BN; idx += BLOCK) {
            int kk = idx / BN, nn = idx % BN;
            int gk = k0 + kk, gn = n0 + nn;
            Bs[kk][nn] = (gk < K && gn < N) ? B[(size_t)gk * ldb + gn] : 0;
        }
        __syncthreads();
        for (int kk = 0; kk < BK; ++kk) {
            // pack the 2x2 products of this thread into one SWAR word: bytes (a0b0, a0b1, a1b0, a1b1)
            uint32_t a0 = As[kk][ty], a1 = As[kk][ty + 16], b0 = Bs[kk][tx], b1 = Bs[kk][tx + 16];
            uint32_t av = a0 | (a0 << 8) | (a1 << 16) | (a1 << 24);
            uint32_t bv = b0 | (b1 << 8) | (b0 << 16) | (b1 << 24);
            acc[0][0] ^= f.mul(av, bv);
        }
        __syncthreads();
    }
    uint32_t r = acc[0][0];
    int gm0 = m0 + ty, gm1 = m0 + ty + 16, gn0 = n0 + tx, gn1 = n0 + tx + 16;
    if (gm0 < M && gn0 < N) C[(size_t)gm0 * ldc + gn0] = (uint8_t)(r & 0xff);
    if (gm0 < M && gn1 < N) C[(size_t)gm0 * ldc + gn1] = (uint8_t)((r >> 8) & 0xff);
    if (gm1 < M && gn0 < N) C[(size_t)gm1 * ldc + gn0] = (uint8_t)((r >> 16) & 0xff);
    if (gm1 < M && gn1 < N) C[(size_t)gm1 * ldc + gn1] = (uint8_t)(r >> 24);
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


template <class F>
__global__ __launch_bounds__(BLOCK) void k_splitk_sum(F f, const typename F::elem* __restrict__ part, int KS, int M, int N,
                                                       typename F::elem* __restrict__ C, size_t ldc) {
    const size_t idx = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    if (idx >= (size_t)M * N) return;
    typedef typename F::word W;
    const size_t mn = (size_t)M * N;
    W r = ld_elem<F>(part, idx);
    int s = 1;
    for (; s + 4 <= KS; s += 4) {                     // four loads in flight
        const W t0 = ld_elem<F>(part, (size_t)s * mn + idx), t1 = ld_elem<F>(part, (size_t)(s + 1) * mn + idx);
        const W t2 = ld_elem<F>(part, (size_t)(s + 2) * mn + idx), t3 = ld_elem<F>(part, (size_t)(s + 3) * mn + idx);
        r = f.add(f.add(r, f.add(t0, t1)), f.add(t2, t3));
    }
    for (; s < KS; ++s) r = f.add(r, ld_elem<F>(part, (size_t)s * mn + idx));
    st_elem<F>(C, (idx / N) * ldc + idx % N, r);
}


// smallest M*N*K that goes to the matrix cores (FFGPU_MM_MFMA_MIN overrides; tuned with tools/mm_threshold.py)
inline double mfma_min_macs() {
    static double v = -1;
    if (v < 0) {
        const char* e = getenv("FFGPU_MM_MFMA_MIN");
        v = e ? atof(e) : 8e7;
    }
    return v;
}

// ---- dense product on the int8 matrix cores ---------------------------------------------------------------
// An exact modular GEMM as integer GEMMs of signed 8-bit DIGITS.  Every operand is first replaced by a
// representative x' = x or x - p (congruent mod p) that has exactly L = 8 (4 for 32-bit storage) base-256 digits
// d_l in [-128, 127] (limb_digits); then
//     (A B)[i][j] = sum_d 256^d D_d[i][j],   D_d = sum_{la + lb = d} A_la B_lb     (2L - 1 integer matrices)
// and every D_d is accumulated by v_mfma_i32_32x32x32_i8 in an i32 accumulator (L * 128^2 * K < 2^31 for a K chunk
// of 8192).  The epilogue evaluates the signed sum by Horner in the field (y * 256 +- |D_d|: muladd_small) -- the
// only place the modulus enters -- so the result is bit-identical to the reduce-once object matmul of
// finfields.py:1126-1135.  This is the one GEMM-shaped piece of the path and the only use of MFMA here: 64 int8
// MFMAs per 64-bit multiply-accumulate still beat 4 quarter-rate v_mad_u64_u32 several times over.
// Operand digits are int8 with k contiguous in runs of 16 (a lane's MFMA fragment -- one row, 16 consecutive k -- is
// one 16-byte load), A by rows and B TRANSPOSED (by columns), zero padded to multiples of 64 rows / 32 k and tiled
// per (64-row block, k-step) as described at limb_off below.
// One wave = one 32x32 output tile with all 2L-1 accumulators (240 registers for L = 8) resident in the
// accumulator half of the register file; 4 waves per workgroup (64x64).
typedef int ff_v4i __attribute__((ext_vector_type(4)));
typedef int ff_v16i __attribute__((ext_vector_type(16)));
enum { LIMB_KCHUNK = 8192 };

// Digit-plane layout: TILED so that what a workgroup fetches per k-step is contiguous.  The digits of a 64-row block
// for one 32-wide k-step form one block of L x 2 KiB, [digit l][k half][row][16 k] -- byte for byte the LDS image
// of the tile -- so the 256 threads of a workgroup read it as consecutive 16-byte chunks (1 KiB per wave
// instruction).  With plain row-major planes [l][row][k] the same fetch touches a different 128-byte line in every
// lane and each line is re-fetched from L2 for four k-steps: the fetch cost 37 % of the kernel (measured by
// switching it off).  Rows are padded to multiples of 64, k to multiples of 32.
template <int L>
__host__ __device__ __forceinline__ size_t limb_off(int l, int row, int k, int Kp) {
    return ((((size_t)(row >> 6) * (size_t)(Kp >> 5) + (size_t)(k >> 5)) * L + l) << 11) + (size_t)(((k >> 4) & 1) << 10) +
           (size_t)((row & 63) << 4) + (size_t)(k & 15);
}

// Epilogue of the 8-digit product: sum_d 256^d D_d mod p for the 15 signed diagonal sums |D_d| <= 2^30 of one
// output.  Horner in the field costs a modular multiply-add and a sign fix per diagonal (~800 instructions per
// output: a quarter of the kernel's time, with one wave per SIMD nothing overlaps it).  Instead the sum is formed
// as an exact INTEGER first -- diagonals 4 apart are 32 bits apart, so
//     lo_r = D_r + 2^32 D_{r+4},  hi_r = D_{r+8} + 2^32 D_{r+12}   (int64, r = 0..3)
//     V = Lo + 2^64 Hi,  Lo = sum_r 2^(8r) lo_r,  Hi = sum_r 2^(8r) hi_r   (|.| < 2^88: __int128)
// -- and reduced once: X mod p = (X mod 2^64) + (2^64 mod p) * (X >> 64) with the small signed high part.
template <class F>
__device__ __forceinline__ typename F::word limb_signed(const F& f, int64_t v) {
    typedef typename F::word W;
    const W w = f.reduce_raw((W)(uint64_t)(v < 0 ? -v : v));
    return v < 0 ? f.neg(w) : w;
}
template <class F>
__device__ __forceinline__ typename F::word limb_red128(const F& f, __int128 x, typename F::word r64) {
    typedef typename F::word W;
    return f.add(f.reduce_raw((W)(uint64_t)x), f.mul(r64, limb_signed(f, (int64_t)(x >> 64))));
}
template <class F>
__device__ __forceinline__ typename F::word limb_combine15(const F& f, const int (&d)[15], typename F::word r64) {
    __int128 lo = 0, hi = 0;
#pragma unroll
    for (int r = 3; r >= 0; --r) {
        const int64_t lr = (int64_t)d[r] + ((int64_t)d[r + 4] << 32);
        const int64_t hr = (int64_t)d[r + 8] + (r + 12 < 15 ? ((int64_t)d[r + 12] << 32) : (int64_t)0);
        lo = (lo << 8) + (__int128)lr;
        hi = (hi << 8) + (__int128)hr;
    }
    // V = (lo mod 2^64) + 2^64 T,  T = hi + (lo >> 64)  (|T| < 2^89)
    const __int128 t = hi + (lo >> 64);
    return f.add(f.reduce_raw((typename F::word)(uint64_t)lo), f.mul(r64, limb_red128(f, t, r64)));
}

template <class F, int L>
__global__ __launch_bounds__(BLOCK) void k_limb_split_a(const typename F::elem* __restrict__ A, size_t lda, uint64_t p,
                                                         int8_t* __restrict__ Ap, int M, int K, int Mp, int Kp) {
    const size_t idx = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    if (idx >= (size_t)Mp * Kp) return;
    const int row = (int)(idx / Kp), kk = (int)(idx % Kp);
    uint64_t v = 0;
    if (row < M && kk < K) v = (uint64_t)ld_elem<F>(A, (size_t)row * lda + kk);
    int8_t d[L];
    limb_digits<L>(v, p, d);
#pragma unroll
    for (int l = 0; l < L; ++l) Ap[limb_off<L>(l, row, kk, Kp)] = d[l];
}
// B (K x N, leading dimension ldb) -> planes [l][Np][Kp] through a 32x32 LDS tile (coalesced reads and writes)
template <class F, int L>
__global__ __launch_bounds__(BLOCK) void k_limb_split_bt(const typename F::elem* __restrict__ B, size_t ldb, uint64_t p,
                                                          int8_t* __restrict__ Bp, int K, int N, int Np, int Kp) {
    __shared__ uint64_t tile[32][33];
    const int n0 = blockIdx.x * 32, k0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int r = ty; r < 32; r += 8) {
        const int kk = k0 + r, nn = n0 + tx;
        tile[r][tx] = (kk < K && nn < N) ? (uint64_t)ld_elem<F>(B, (size_t)kk * ldb + nn) : 0;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {                       // r: column n of the tile, tx: k
        int8_t d[L];
        limb_digits<L>(tile[tx][r], p, d);
#pragma unroll
        for (int l = 0; l < L; ++l) Bp[limb_off<L>(l, n0 + r, k0 + tx, Kp)] = d[l];
    }
}

template <class F, int L>
__global__ __launch_bounds__(BLOCK) void k_limb_gemm(F f, const int8_t* __restrict__ Ap, const int8_t* __restrict__ Bp,
                                                      typename F::elem* __restrict__ C, size_t ldc, int M, int N, int Mp,
                                                      int Np, int Kp, int kb, int ke, int accumulate) {
    typedef typename F::word W;
    constexpr int ND = 2 * L - 1;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int m0 = blockIdx.y * 64 + (wave >> 1) * 32, n0 = blockIdx.x * 64 + (wave & 1) * 32;
    const int r = lane & 31, h = lane >> 5;
    ff_v16i acc[ND];
#pragma unroll
    for (int d = 0; d < ND; ++d) acc[d] = (ff_v16i){0};
    auto pa = [&](int l, int k) { return Ap + limb_off<L>(l, m0 + r, k + 16 * h, Kp); };
    auto pb = [&](int l, int k) { return Bp + limb_off<L>(l, n0 + r, k + 16 * h, Kp); };
    // software pipeline: with all 2L-1 accumulators resident there is ONE wave per SIMD, so nothing else hides the
    // latency of the fragment loads.  The B fragments of step k+1 are fetched into a second set of registers
    // before the L*L MFMAs of step k; an A fragment is dead after its row of MFMAs and is refilled in place.
    ff_v4i a[L], b[L], bn[L];
#pragma unroll
    for (int l = 0; l < L; ++l) {
        a[l] = *reinterpret_cast<const ff_v4i*>(pa(l, kb));
        b[l] = *reinterpret_cast<const ff_v4i*>(pb(l, kb));
    }
    for (int k0 = kb; k0 < ke; k0 += 32) {
        const int kn = k0 + 32 < ke ? k0 + 32 : k0;          // last step: harmless reload of the same fragments
#pragma unroll
        for (int l = 0; l < L; ++l) bn[l] = *reinterpret_cast<const ff_v4i*>(pb(l, kn));
#pragma unroll
        for (int la = 0; la < L; ++la) {
#pragma unroll
            for (int lb = 0; lb < L; ++lb)
                acc[la + lb] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[la], b[lb], acc[la + lb], 0, 0, 0);
            a[la] = *reinterpret_cast<const ff_v4i*>(pa(la, kn));
        }
#pragma unroll
        for (int l = 0; l < L; ++l) b[l] = bn[l];
    }
    // epilogue: sum_d 256^d D_d mod p, signed digits sums: Horner from the top diagonal, one diagonal at a time for
    // all 16 results of the lane
    W res[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const int dv = acc[ND - 1][q];
        const W w = f.reduce_raw((W)(uint32_t)(dv < 0 ? -dv : dv));
        res[q] = dv < 0 ? f.neg(w) : w;
    }
#pragma unroll
    for (int d = ND - 2; d >= 0; --d)
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int dv = acc[d][q];
            const W w = f.reduce_raw((W)(uint32_t)(dv < 0 ? -dv : dv));
            res[q] = f.muladd_small(res[q], 256u, dv < 0 ? f.neg(w) : w);
        }
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const int col = n0 + (lane & 31), row = m0 + (q & 3) + 8 * (q >> 2) + 4 * (lane >> 5);
        if (row < M && col < N) {
            W v = res[q];
            if (accumulate) v = f.add(v, ld_elem<F>(C, (size_t)row * ldc + col));
            st_elem<F>(C, (size_t)row * ldc + col, v);
        }
    }
}

// The same product with the operand tiles staged through LDS: the four waves of a workgroup (64x64 outputs)
// share one copy of the 64-row A tile and the 64-column B tile per k-step (2 x L x 2 KiB, double buffered), which
// halves the L2 -> CU traffic that bounds the direct-load variant.  LDS layout [plane][k-half][row][16 bytes]:
// the 16 lanes a ds_read_b128 phase serves read 256 contiguous bytes (conflict-free).  (A variant that also
// double-buffers the FRAGMENT registers -- LDS reads of step s+1 issued before the MFMAs of step s -- measured the
// same: the LDS latency is not what the loop waits for; profiles/r02_limb_gemm.md.)
// BRAW: the right operand is the field matrix itself (row-major K x N, leading dimension ldb) instead of digit
// planes: a thread fetches eight consecutive k of one column (the loads of a wave cover 512 contiguous bytes per
// k), converts them to signed digits in registers (limb_digits_packed: three instructions per element), transposes
// the 8 x 8 digit bytes with v_perm_b32 and writes eight 8-byte runs into the same LDS image.  For a few rows
// against a big matrix (a batch of activations times a weight matrix: M <= 128) this removes the pass that writes and
// re-reads the planes of B, which cost more than a third of the product (64 x 4096 x 4096: split 54-67 us, product 94).
__device__ __forceinline__ void transpose4x4_bytes(uint32_t r0, uint32_t r1, uint32_t r2, uint32_t r3, uint32_t (&o)[4]) {
    const uint32_t t0 = __builtin_amdgcn_perm(r1, r0, 0x05010400u), t1 = __builtin_amdgcn_perm(r1, r0, 0x07030602u);
    const uint32_t t2 = __builtin_amdgcn_perm(r3, r2, 0x05010400u), t3 = __builtin_amdgcn_perm(r3, r2, 0x07030602u);
    o[0] = __builtin_amdgcn_perm(t2, t0, 0x05040100u);
    o[1] = __builtin_amdgcn_perm(t2, t0, 0x07060302u);
    o[2] = __builtin_amdgcn_perm(t3, t1, 0x05040100u);
    o[3] = __builtin_amdgcn_perm(t3, t1, 0x07060302u);
}

template <class F, int L, bool BRAW = false>
__global__ __launch_bounds__(BLOCK) void k_limb_gemm_lds(F f, const int8_t* __restrict__ Ap, const int8_t* __restrict__ Bp,
                                                          typename F::elem* __restrict__ C, size_t ldc, int M, int N, int Mp,
                                                          int Np, int Kp, int kb, int ke, int accumulate, int kslice,
                                                          size_t zstride, const typename F::elem* __restrict__ Braw = nullptr,
                                                          size_t ldb = 0, int K = 0, uint64_t pmod = 0) {
    static_assert(!BRAW || (L == 8 && sizeof(typename F::elem) == 8), "raw right operand: 64-bit storage, eight digits");
    typedef typename F::word W;
    constexpr int ND = 2 * L - 1;
    constexpr int CHUNKS = L * 2 * 64;                 // 16-byte chunks of one operand tile per k-step
    constexpr int PER_THREAD = CHUNKS / BLOCK;         // = L / 2 (L is 4 or 8)
    static_assert(CHUNKS % BLOCK == 0, "tile chunks must divide evenly over the workgroup");
    __shared__ ff_v4i sA[2][CHUNKS];
    __shared__ ff_v4i sB[2][CHUNKS];
    if (kslice > 0) {                                  // split-K: slice blockIdx.z -> its own slab of C
        kb += blockIdx.z * kslice;
        ke = kb + kslice < ke ? kb + kslice : ke;
        C += (size_t)blockIdx.z * zstride;
        if (kb >= ke) { kb = 0; ke = 0; }              // empty slice: writes zeros
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = (wave >> 1) * 32, wn = (wave & 1) * 32;
    const int bm0 = blockIdx.y * 64, bn0 = blockIdx.x * 64;
    const int r = lane & 31, h = lane >> 5;
    ff_v16i acc[ND];
#pragma unroll
    for (int d = 0; d < ND; ++d) acc[d] = (ff_v16i){0};
    // chunk c of a tile: plane l = c / 128, k-half hh = (c / 64) % 2, row = c % 64  (== its LDS index)
    ff_v4i ga[PER_THREAD], gb[BRAW ? 1 : PER_THREAD];
    uint64_t braw[BRAW ? 8 : 1];
    const int bcol = threadIdx.x & 63, bkg = threadIdx.x >> 6;      // BRAW: this thread's column and group of eight k
    auto fetch = [&](int k0) {
#pragma unroll
        for (int u = 0; u < PER_THREAD; ++u) {
            const int c = threadIdx.x + u * BLOCK;
            const int l = c >> 7, hh = (c >> 6) & 1, row = c & 63;
            ga[u] = *reinterpret_cast<const ff_v4i*>(Ap + limb_off<L>(l, bm0 + row, k0 + 16 * hh, Kp));
            if constexpr (!BRAW) gb[u] = *reinterpret_cast<const ff_v4i*>(Bp + limb_off<L>(l, bn0 + row, k0 + 16 * hh, Kp));
        }
        if constexpr (BRAW) {
            const int col = bn0 + bcol;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int kk = k0 + bkg * 8 + i;
                braw[i] = (kk < K && col < N) ? (uint64_t)Braw[(size_t)kk * ldb + col] : 0;
            }
        }
    };
    auto stash = [&](int buf) {
#pragma unroll
        for (int u = 0; u < PER_THREAD; ++u) {
            sA[buf][threadIdx.x + u * BLOCK] = ga[u];
            if constexpr (!BRAW) sB[buf][threadIdx.x + u * BLOCK] = gb[u];
        }
        if constexpr (BRAW) {
            uint32_t lo[8], hi[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const uint64_t d = limb_digits_packed<8>(braw[i], pmod);      // byte l = digit l of element k
                lo[i] = (uint32_t)d;
                hi[i] = (uint32_t)(d >> 32);
            }
            uint32_t w[4][4];                      // [digit quad: lo/k0-3, lo/k4-7, hi/k0-3, hi/k4-7][digit in the quad]
            transpose4x4_bytes(lo[0], lo[1], lo[2], lo[3], w[0]);
            transpose4x4_bytes(lo[4], lo[5], lo[6], lo[7], w[1]);
            transpose4x4_bytes(hi[0], hi[1], hi[2], hi[3], w[2]);
            transpose4x4_bytes(hi[4], hi[5], hi[6], hi[7], w[3]);
            // digit plane l, k-half hh, column: 16-byte chunk ((l * 2 + hh) * 64 + col); this thread's eight k are its
            // lower or upper eight bytes
            uint64_t* sb8 = reinterpret_cast<uint64_t*>(&sB[buf][0]);
            const int hh = bkg >> 1, half = bkg & 1;
#pragma unroll
            for (int l = 0; l < 8; ++l) {
                const uint64_t run = (uint64_t)w[(l >> 2) * 2][l & 3] | ((uint64_t)w[(l >> 2) * 2 + 1][l & 3] << 32);
                sb8[(((l * 2 + hh) * 64 + bcol) << 1) + half] = run;
            }
        }
    };
    if (kb < ke) {
        fetch(kb);
        stash(0);
    }
    __syncthreads();
    int cur = 0;
    for (int k0 = kb; k0 < ke; k0 += 32) {
        const bool more = k0 + 32 < ke;
        if (more) fetch(k0 + 32);                      // in flight during the MFMAs below
        ff_v4i a[L], b[L];
#pragma unroll
        for (int l = 0; l < L; ++l) {
            a[l] = sA[cur][(l * 2 + h) * 64 + wm + r];
            b[l] = sB[cur][(l * 2 + h) * 64 + wn + r];
        }
#pragma unroll
        for (int la = 0; la < L; ++la)
#pragma unroll
            for (int lb = 0; lb < L; ++lb)
                acc[la + lb] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[la], b[lb], acc[la + lb], 0, 0, 0);
        if (more) stash(cur ^ 1);
        __syncthreads();
        cur ^= 1;
    }
    W res[16];
    if constexpr (L == 8 && sizeof(W) == 8) {
        const W t32 = f.reduce_raw((W)(1ull << 32));
        const W r64 = f.mul(t32, t32);                 // 2^64 mod p
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            int dq[15];
#pragma unroll
            for (int d = 0; d < 15; ++d) dq[d] = acc[d][q];
            res[q] = limb_combine15(f, dq, r64);
        }
    } else {
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int dv = acc[ND - 1][q];
            const W w = f.reduce_raw((W)(uint32_t)(dv < 0 ? -dv : dv));
            res[q] = dv < 0 ? f.neg(w) : w;
        }
#pragma unroll
        for (int d = ND - 2; d >= 0; --d)
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int dv = acc[d][q];
                const W w = f.reduce_raw((W)(uint32_t)(dv < 0 ? -dv : dv));
                res[q] = f.muladd_small(res[q], 256u, dv < 0 ? f.neg(w) : w);
            }
    }
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const int col = bn0 + wn + (lane & 31), row = bm0 + wm + (q & 3) + 8 * (q >> 2) + 4 * (lane >> 5);
        if (row < M && col < N) {
            W v = res[q];
            if (accumulate) v = f.add(v, ld_elem<F>(C, (size_t)row * ldc + col));
            st_elem<F>(C, (size_t)row * ldc + col, v);
        }
    }
}

// ---- the same product with the operand tiles streamed STRAIGHT into LDS, three stages deep (round 4) ------------
// k_limb_gemm_lds holds one tile ahead in registers: its fetch is issued one k-step (64 MFMAs = ~1 us) before the
// data is needed, which does not cover the HBM / L2 latency at one wave per SIMD (the accumulators take the register
// file: 512 of 512) -- measured: the operand fetch cost 37 % of the 4096^3 product, and the 64-row shape ran at 110 us
// against 31 us of MFMA work.  Here the digit-plane tiles (byte for byte their LDS image: limb_off) and, for BRAW, the
// raw rows of B go global -> LDS without passing through registers (global_load_lds_dwordx4: LDS address = wave-uniform
// base + lane * 16), TWO tiles ahead in a ring of three stages; the waves wait with a COUNTED vmcnt (the tile issued
// last stays in flight across the barrier) and synchronise with raw s_barrier (a __syncthreads would drain vmcnt to
// 0).  BRAW: the raw 64-bit elements of tile s + 1 are converted to digit bytes LDS -> LDS (same arithmetic as the
// register variant) while the MFMAs of tile s run.  No staging registers (32 fewer).  Requires K padded to 32 (planes:
// always) and, for BRAW, K % 32 == 0, N % 64 == 0 and 16-byte aligned rows of B; the launcher falls back to
// k_limb_gemm_lds otherwise.  FFGPU_MM_GLDS=0 selects the register-staged kernel (A/B measurements, parity tests).
template <class F>
__device__ __forceinline__ void limb_epilogue8(const F& f, const ff_v16i (&acc)[15], typename F::elem* __restrict__ C, size_t ldc,
                                               int M, int N, int bm0, int bn0, int wm, int wn, int lane, int accumulate) {
    typedef typename F::word W;
    const W t32 = f.reduce_raw((W)(1ull << 32));
    const W r64 = f.mul(t32, t32);                 // 2^64 mod p
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        int dq[15];
#pragma unroll
        for (int d = 0; d < 15; ++d) dq[d] = acc[d][q];
        W v = limb_combine15(f, dq, r64);
        const int col = bn0 + wn + (lane & 31), row = bm0 + wm + (q & 3) + 8 * (q >> 2) + 4 * (lane >> 5);
        if (row < M && col < N) {
            if (accumulate) v = f.add(v, ld_elem<F>(C, (size_t)row * ldc + col));
            st_elem<F>(C, (size_t)row * ldc + col, v);
        }
    }
}

enum { GLDS_TILE = 8 * 2 * 64 * 16 };              // bytes of one operand tile per k-step (L = 8): 16 KiB
template <class F, bool BRAW>
__global__ __launch_bounds__(BLOCK) void k_limb_gemm_glds(F f, const int8_t* __restrict__ Ap, const int8_t* __restrict__ Bp,
                                                           typename F::elem* __restrict__ C, size_t ldc, int M, int N, int Kp,
                                                           int kb, int ke, int accumulate, int kslice, size_t zstride,
                                                           const typename F::elem* __restrict__ Braw, size_t ldb, uint64_t pmod) {
    static_assert(sizeof(typename F::elem) == 8, "eight digits, 64-bit storage");
    constexpr int L = 8;
    // LDS: A stages [3][16 KiB]; planes of B: stages [3][16 KiB]; BRAW: raw stages [3][32 k][64 columns] uint64 + digit tiles [2][16 KiB]
    extern __shared__ __attribute__((aligned(16))) unsigned char glds_smem[];
    unsigned char* sA = glds_smem;
    unsigned char* sBst = glds_smem + 3 * GLDS_TILE;                   // planes of B, or the raw stages
    unsigned char* sBd = glds_smem + 6 * GLDS_TILE;                    // BRAW only: converted digit tiles [2]
    if (kslice > 0) {                                  // split-K: slice blockIdx.z -> its own slab of C
        kb += blockIdx.z * kslice;
        ke = kb + kslice < ke ? kb + kslice : ke;
        C += (size_t)blockIdx.z * zstride;
        if (kb >= ke) { kb = 0; ke = 0; }              // empty slice: writes zeros
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = (wave >> 1) * 32, wn = (wave & 1) * 32;
    const int bm0 = blockIdx.y * 64, bn0 = blockIdx.x * 64;
    const int r = lane & 31, h = lane >> 5;
    ff_v16i acc[15];
#pragma unroll
    for (int d = 0; d < 15; ++d) acc[d] = (ff_v16i){0};
    const int nsteps = (ke - kb) >> 5;
    typedef __attribute__((address_space(3))) void lds_void;
    // tile `s` -> stage s % 3: four 1 KiB pieces per wave and operand
    auto issue = [&](int s_) {
        const int k0 = kb + 32 * s_, st = s_ % 3;
        const int8_t* at = Ap + limb_off<L>(0, bm0, k0, Kp);           // 16 KiB contiguous, already in LDS order
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int piece = wave * 4 + u;                            // 1 KiB pieces 0..15
            __builtin_amdgcn_global_load_lds(at + piece * 1024 + lane * 16, (lds_void*)(sA + st * GLDS_TILE + piece * 1024), 16, 0, 0);
        }
        if constexpr (!BRAW) {
            const int8_t* bt = Bp + limb_off<L>(0, bn0, k0, Kp);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int piece = wave * 4 + u;
                __builtin_amdgcn_global_load_lds(bt + piece * 1024 + lane * 16, (lds_void*)(sBst + st * GLDS_TILE + piece * 1024), 16, 0, 0);
            }
        } else {
            // raw rows k0 .. k0 + 31, 64 columns of 8 bytes: one instruction = two k rows (lanes 0..31 / 32..63, two columns each)
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int kr = (wave * 4 + u) * 2;
                const int8_t* src = reinterpret_cast<const int8_t*>(Braw + (size_t)(k0 + kr + (lane >> 5)) * ldb + bn0 + (lane & 31) * 2);
                __builtin_amdgcn_global_load_lds(src, (lds_void*)(sBst + st * GLDS_TILE + kr * 512), 16, 0, 0);
            }
        }
    };
    // BRAW: raw stage of tile s -> digit tile s & 1 (thread: column bcol, eight consecutive k)
    const int bcol = threadIdx.x & 63, bkg = threadIdx.x >> 6;
    auto convert = [&](int s_) {
        const uint64_t* raw = reinterpret_cast<const uint64_t*>(sBst + (s_ % 3) * GLDS_TILE);
        uint32_t lo[8], hi[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const uint64_t d = limb_digits_packed<8>(raw[(bkg * 8 + i) * 64 + bcol], pmod);      // byte l = digit l of element k
            lo[i] = (uint32_t)d;
            hi[i] = (uint32_t)(d >> 32);
        }
        uint32_t w[4][4];
        transpose4x4_bytes(lo[0], lo[1], lo[2], lo[3], w[0]);
        transpose4x4_bytes(lo[4], lo[5], lo[6], lo[7], w[1]);
        transpose4x4_bytes(hi[0], hi[1], hi[2], hi[3], w[2]);
        transpose4x4_bytes(hi[4], hi[5], hi[6], hi[7], w[3]);
        uint64_t* sb8 = reinterpret_cast<uint64_t*>(sBd + (s_ & 1) * GLDS_TILE);
        const int hh = bkg >> 1, half = bkg & 1;
#pragma unroll
        for (int l = 0; l < 8; ++l) {
            const uint64_t run = (uint64_t)w[(l >> 2) * 2][l & 3] | ((uint64_t)w[(l >> 2) * 2 + 1][l & 3] << 32);
            sb8[(((l * 2 + hh) * 64 + bcol) << 1) + half] = run;
        }
    };
    auto barrier = [&]() {                             // LDS traffic of this wave done, then the workgroup meets (vmcnt untouched)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };
    constexpr int PER_TILE = 8;                        // global_load_lds instructions per wave and tile (4 for A + 4 for B)
    if (nsteps > 0) {
        issue(0);
        if (nsteps > 1) {
            issue(1);
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        barrier();                                     // tile 0 is in LDS for every wave
        if constexpr (BRAW) {
            convert(0);
            barrier();
        }
    }
    static_assert(PER_TILE == 8, "the counted waits below leave exactly one tile (8 instructions) in flight");
    for (int s_ = 0; s_ < nsteps; ++s_) {
        // stage (s + 2) % 3 held tile s - 1: its last readers (the MFMAs of step s - 1, the conversion in step s - 2)
        // finished before the barrier that ended step s - 1
        if (s_ + 2 < nsteps) issue(s_ + 2);
        const ff_v4i* a4 = reinterpret_cast<const ff_v4i*>(sA + (s_ % 3) * GLDS_TILE);
        const ff_v4i* b4 = reinterpret_cast<const ff_v4i*>(BRAW ? sBd + (s_ & 1) * GLDS_TILE : sBst + (s_ % 3) * GLDS_TILE);
        ff_v4i a[L], b[L];
#pragma unroll
        for (int l = 0; l < L; ++l) {
            a[l] = a4[(l * 2 + h) * 64 + wm + r];
            b[l] = b4[(l * 2 + h) * 64 + wn + r];
        }
#pragma unroll
        for (int la = 0; la < L; ++la)
#pragma unroll
            for (int lb = 0; lb < L; ++lb)
                acc[la + lb] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[la], b[lb], acc[la + lb], 0, 0, 0);
        if (s_ + 1 < nsteps) {
            // tile s + 1 (issued a whole step ago) must have landed; tile s + 2 stays in flight
            if (s_ + 2 < nsteps) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            barrier();
            if constexpr (BRAW) {
                convert(s_ + 1);
                barrier();
            }
        }
    }
    limb_epilogue8(f, acc, C, ldc, M, N, bm0, bn0, wm, wn, lane, accumulate);
}

// ---- the matrix-core product for primes of 65..128 bits ---------------------------------------------------------
// L = 12 digits (96-bit storage) or 16: 2L-1 = 23 / 31 diagonals do not fit the register file at once, so the
// product runs in PASSES over ranges of diagonals [D0, D0+NDP): a pass issues only the MFMAs whose digit pair lies
// on its diagonals (the work adds up to L^2 per k-step over all passes), evaluates its part by Horner and adds
// 256^D0 times it to C.  K chunks of 4096 keep the i32 accumulators exact (16 * 128^2 * 4096 = 2^30).
enum { LIMB_KCHUNK_WIDE = 4096 };

template <class F, int L>
__global__ __launch_bounds__(BLOCK) void k_limb_split_a_wide(const typename F::elem* __restrict__ A, size_t lda, uint64_t plo,
                                                              uint64_t phi, int8_t* __restrict__ Ap, int M, int K, int Mp,
                                                              int Kp) {
    const size_t idx = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    if (idx >= (size_t)Mp * Kp) return;
    const int row = (int)(idx / Kp), kk = (int)(idx % Kp);
    uint64_t lo = 0, hi = 0;
    if (row < M && kk < K) {
        const typename F::word w = ld_elem<F>(A, (size_t)row * lda + kk);
        lo = w.lo;
        hi = w.hi;
    }
    int8_t d[L];
    limb_digits_wide<L>(lo, hi, plo, phi, d);
#pragma unroll
    for (int l = 0; l < L; ++l) Ap[limb_off<L>(l, row, kk, Kp)] = d[l];
}
template <class F, int L>
__global__ __launch_bounds__(BLOCK) void k_limb_split_bt_wide(const typename F::elem* __restrict__ B, size_t ldb, uint64_t plo,
                                                               uint64_t phi, int8_t* __restrict__ Bp, int K, int N, int Np,
                                                               int Kp) {
    __shared__ uint64_t tlo[32][33];
    __shared__ uint64_t thi[32][33];
    const int n0 = blockIdx.x * 32, k0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int r = ty; r < 32; r += 8) {
        const int kk = k0 + r, nn = n0 + tx;
        uint64_t lo = 0, hi = 0;
        if (kk < K && nn < N) {
            const typename F::word w = ld_elem<F>(B, (size_t)kk * ldb + nn);
            lo = w.lo;
            hi = w.hi;
        }
        tlo[r][tx] = lo;
        thi[r][tx] = hi;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        int8_t d[L];
        limb_digits_wide<L>(tlo[tx][r], thi[tx][r], plo, phi, d);
#pragma unroll
        for (int l = 0; l < L; ++l) Bp[limb_off<L>(l, n0 + r, k0 + tx, Kp)] = d[l];
    }
}

// one pass: diagonals D0 .. D0+NDP-1; scale = 256^D0 mod p (prepared); accumulate: add to C instead of writing it
template <class F, int L, int D0, int NDP>
__global__ __launch_bounds__(BLOCK) void k_limb_gemm_wide(F f, const int8_t* __restrict__ Ap, const int8_t* __restrict__ Bp,
                                                           typename F::elem* __restrict__ C, size_t ldc, int M, int N, int Mp,
                                                           int Np, int Kp, int kb, int ke, int accumulate,
                                                           typename F::word scale) {
    typedef typename F::word W;
    constexpr int CHUNKS = L * 2 * 64;
    constexpr int PER_THREAD = CHUNKS / BLOCK;
    static_assert(CHUNKS % BLOCK == 0, "tile chunks must divide evenly over the workgroup");
    __shared__ ff_v4i sA[2][CHUNKS];
    __shared__ ff_v4i sB[2][CHUNKS];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = (wave >> 1) * 32, wn = (wave & 1) * 32;
    const int bm0 = blockIdx.y * 64, bn0 = blockIdx.x * 64;
    const int r = lane & 31, h = lane >> 5;
    ff_v16i acc[NDP];
#pragma unroll
    for (int d = 0; d < NDP; ++d) acc[d] = (ff_v16i){0};
    ff_v4i ga[PER_THREAD], gb[PER_THREAD];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int u = 0; u < PER_THREAD; ++u) {
            const int c = threadIdx.x + u * BLOCK;
            const int l = c >> 7, hh = (c >> 6) & 1, row = c & 63;
            ga[u] = *reinterpret_cast<const ff_v4i*>(Ap + limb_off<L>(l, bm0 + row, k0 + 16 * hh, Kp));
            gb[u] = *reinterpret_cast<const ff_v4i*>(Bp + limb_off<L>(l, bn0 + row, k0 + 16 * hh, Kp));
        }
    };
    auto stash = [&](int buf) {
#pragma unroll
        for (int u = 0; u < PER_THREAD; ++u) {
            sA[buf][threadIdx.x + u * BLOCK] = ga[u];
            sB[buf][threadIdx.x + u * BLOCK] = gb[u];
        }
    };
    if (kb < ke) {
        fetch(kb);
        stash(0);
    }
    __syncthreads();
    int cur = 0;
    for (int k0 = kb; k0 < ke; k0 += 32) {
        const bool more = k0 + 32 < ke;
        if (more) fetch(k0 + 32);
        ff_v4i a[L], b[L];
#pragma unroll
        for (int l = 0; l < L; ++l) {
            a[l] = sA[cur][(l * 2 + h) * 64 + wm + r];
            b[l] = sB[cur][(l * 2 + h) * 64 + wn + r];
        }
#pragma unroll
        for (int la = 0; la < L; ++la)
#pragma unroll
            for (int lb = 0; lb < L; ++lb)
                if (la + lb >= D0 && la + lb < D0 + NDP)
                    acc[la + lb - D0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[la], b[lb], acc[la + lb - D0], 0, 0, 0);
        if (more) stash(cur ^ 1);
        __syncthreads();
        cur ^= 1;
    }
    auto to_field = [&](int dv) -> W {
        W w;
        w.lo = (uint64_t)(uint32_t)(dv < 0 ? -dv : dv);
        w.hi = 0;
        w = f.reduce_raw(w);
        return dv < 0 ? f.neg(w) : w;
    };
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const int col = bn0 + wn + (lane & 31), row = bm0 + wm + (q & 3) + 8 * (q >> 2) + 4 * (lane >> 5);
        if (row < M && col < N) {
            W res = to_field(acc[NDP - 1][q]);
#pragma unroll
            for (int d = NDP - 2; d >= 0; --d) res = f.muladd_small(res, 256u, to_field(acc[d][q]));
            if (D0 > 0) res = f.mul(res, scale);
            if (accumulate) res = f.add(res, ld_elem<F>(C, (size_t)row * ldc + col));
            st_elem<F>(C, (size_t)row * ldc + col, res);
        }
    }
}

// ---- skinny products: matrix x few columns, few rows x matrix ---------------------------------------------
// The tiled k_matmul needs both output dimensions to fill the chip; the shapes MPyC's author flags as the
// bottleneck (demos/np_bnnmnist.py:10-15: `L @ W` with a 1 x 4096 activation row and a 4096 x 4096 weight
// matrix, finfields.py:1126-1135) have one output dimension of 1..8.  Both are HBM-bound: the big operand is
// read exactly once, coalesced, the small one stays in L2; products are accumulated unreduced (flush every
// 192 terms, as k_dot_partial) and reduced once.
enum { SKINNY_MAX = 8, SKINNY_FLUSH = 192 };

// C (M x N) = A (M x K) @ B (K x N), N <= SKINNY_MAX: one workgroup per row of A
template <class F, int NN>
__global__ __launch_bounds__(BLOCK) void k_matvec_rows(F f, const typename F::elem* __restrict__ A, size_t lda,
                                                        const typename F::elem* __restrict__ B, size_t ldb,
                                                        typename F::elem* __restrict__ C, size_t ldc, int K, int N,
                                                        int vec, int bvec) {
    typedef Pack<typename F::word> P;
    typedef typename MemPack<F>::type MP;
    typedef typename F::word W;
    __shared__ W sm[BLOCK];
    const size_t row = blockIdx.x;
    const typename F::elem* __restrict__ a = A + row * lda;
    typename F::acc acc[NN];
    W total[NN];
    bool have = false;
    int cnt = 0;
#pragma unroll
    for (int j = 0; j < NN; ++j) f.acc_zero(acc[j]);
    auto flush = [&]() {
#pragma unroll
        for (int j = 0; j < NN; ++j) {
            W part = f.acc_reduce(acc[j]);
            total[j] = have ? f.add(total[j], part) : part;
            f.acc_zero(acc[j]);
        }
        have = true;
        cnt = 0;
    };
    auto term = [&](W x, size_t kk) {
        const W xp = f.prep(x);
        if (bvec) {          // the N values of row kk of B with 16-byte loads (N a multiple of the pack width)
            const MP* __restrict__ br = reinterpret_cast<const MP*>(B + kk * ldb);
#pragma unroll
            for (int jp = 0; jp < (NN + P::N - 1) / P::N; ++jp) {
                if (jp * P::N < N) {
                    const P bp = ldg<false>(br + jp);
#pragma unroll
                    for (int q = 0; q < P::N; ++q)
                        if (jp * P::N + q < NN) f.acc_mac(acc[jp * P::N + q], xp, bp.w[q]);
                }
            }
        } else {
#pragma unroll
            for (int j = 0; j < NN; ++j)
                if (j < N) f.acc_mac(acc[j], xp, ld_elem<F>(B, kk * ldb + j));
        }
        if (++cnt >= SKINNY_FLUSH) flush();
    };
    constexpr int EPV = P::N;
    const int nvec = vec ? K / EPV : 0;
    const MP* __restrict__ av = reinterpret_cast<const MP*>(a);
    for (int i = threadIdx.x; i < nvec; i += BLOCK) {
        const P x = ldg<true>(av + i);
#pragma unroll
        for (int q = 0; q < P::N; ++q) term(x.w[q], (size_t)i * EPV + q);
    }
    for (int kk = nvec * EPV + threadIdx.x; kk < K; kk += BLOCK) term(ld_elem<F>(a, kk), (size_t)kk);
    flush();
#pragma unroll
    for (int j = 0; j < NN; ++j) {
        if (j < N) {
            const W r = block_reduce_add(f, total[j], sm);
            if (threadIdx.x == 0) st_elem<F>(C, row * ldc + j, r);
            __syncthreads();
        }
    }
}

// The same product with R rows of A per workgroup (R = 2 in use): the values of B a thread needs (B is re-read by every workgroup:
// with one row per workgroup the L2 -> CU traffic for B equals the HBM traffic for A) are loaded ONCE per R rows and
// the R row packs are all in flight before the first multiply.  `bpack`: N == 1 with unit-stride, aligned B -- the
// vector is read as 16-byte packs with the same index as A's.
template <class F, int NN, int R>
__global__ __launch_bounds__(BLOCK) void k_matvec_rows_r(F f, const typename F::elem* __restrict__ A, size_t lda,
                                                          const typename F::elem* __restrict__ B, size_t ldb,
                                                          typename F::elem* __restrict__ C, size_t ldc, int M, int K, int N,
                                                          int vec, int bpack) {
    typedef Pack<typename F::word> P;
    typedef typename MemPack<F>::type MP;
    typedef typename F::word W;
    __shared__ W sm[BLOCK];
    const size_t row0 = (size_t)blockIdx.x * R;
    typename F::acc acc[R][NN];
    W total[R][NN];
    bool have = false;
    int cnt = 0;
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int j = 0; j < NN; ++j) f.acc_zero(acc[r][j]);
    auto flush = [&]() {
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int j = 0; j < NN; ++j) {
                W part = f.acc_reduce(acc[r][j]);
                total[r][j] = have ? f.add(total[r][j], part) : part;
                f.acc_zero(acc[r][j]);
            }
        have = true;
        cnt = 0;
    };
    constexpr int EPV = P::N;
    const int nvec = vec ? K / EPV : 0;
    for (int i = threadIdx.x; i < nvec; i += BLOCK) {
        P x[R];
#pragma unroll
        for (int r = 0; r < R; ++r)                                   // rows past M re-read the last row (result discarded)
            x[r] = ldg<true>(reinterpret_cast<const MP*>(A + (row0 + r < (size_t)M ? row0 + r : (size_t)M - 1) * lda) + i);
        W b[EPV][NN];
        if (bpack) {
            const P bp = ldg<false>(reinterpret_cast<const MP*>(B) + i);
#pragma unroll
            for (int q = 0; q < EPV; ++q) b[q][0] = f.prep(bp.w[q]);
        } else {
#pragma unroll
            for (int q = 0; q < EPV; ++q)
#pragma unroll
                for (int j = 0; j < NN; ++j)
                    if (j < N) b[q][j] = f.prep(ld_elem<F>(B, ((size_t)i * EPV + q) * ldb + j));
        }
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int q = 0; q < EPV; ++q)
#pragma unroll
                for (int j = 0; j < NN; ++j)
                    if (j < N) f.acc_mac(acc[r][j], b[q][j], x[r].w[q]);
        cnt += EPV;
        if (cnt >= SKINNY_FLUSH) flush();
    }
    for (int kk = nvec * EPV + threadIdx.x; kk < K; kk += BLOCK) {
#pragma unroll
        for (int j = 0; j < NN; ++j)
            if (j < N) {
                const W bp = f.prep(ld_elem<F>(B, (size_t)kk * ldb + j));
#pragma unroll
                for (int r = 0; r < R; ++r)
                    f.acc_mac(acc[r][j], bp, ld_elem<F>(A, (row0 + r < (size_t)M ? row0 + r : (size_t)M - 1) * lda + kk));
            }
        if (++cnt >= SKINNY_FLUSH) flush();
    }
    flush();
    // R*NN sums over the workgroup: butterfly inside each wave (cross-lane moves, no barrier), then ONE exchange
    // of the per-wave sums through LDS
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int j = 0; j < NN; ++j) {
            W v = total[r][j];
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) v = f.add(v, wave_shfl_xor(v, off));
            if (lane == 0) sm[(r * NN + j) * (BLOCK / 64) + wv] = v;
        }
    __syncthreads();
    if (threadIdx.x < R * NN) {
        const int r = threadIdx.x / NN, j = threadIdx.x % NN;
        W v = sm[threadIdx.x * (BLOCK / 64)];
#pragma unroll
        for (int w2 = 1; w2 < BLOCK / 64; ++w2) v = f.add(v, sm[threadIdx.x * (BLOCK / 64) + w2]);
        if (j < N && row0 + r < (size_t)M) st_elem<F>(C, (row0 + r) * ldc + j, v);
    }
}

// Same product for SHORT rows (K <= 32, many rows: sums over a trailing axis, tall-thin least squares): one
// thread per row -- a row is K contiguous elements, neighbouring threads read neighbouring rows, and B[k][j] is
// wave-uniform (scalar loads).
template <class F, int NN>
__global__ __launch_bounds__(BLOCK) void k_matvec_short_rows(F f, const typename F::elem* __restrict__ A, size_t lda,
                                                              const typename F::elem* __restrict__ B, size_t ldb,
                                                              typename F::elem* __restrict__ C, size_t ldc, int M, int K,
                                                              int N) {
    typedef typename F::word W;
    const size_t gid = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    const size_t gsz = (size_t)gridDim.x * BLOCK;
    for (size_t row = gid; row < (size_t)M; row += gsz) {
        typename F::acc acc[NN];
#pragma unroll
        for (int j = 0; j < NN; ++j) f.acc_zero(acc[j]);
        for (int kk = 0; kk < K; ++kk) {                                  // K <= 32 < SKINNY_FLUSH: no flush needed
            const W x = ld_elem<F>(A, row * lda + kk);
#pragma unroll
            for (int j = 0; j < NN; ++j)
                if (j < N) f.acc_mac(acc[j], f.prep(ld_elem<F>(B, (size_t)kk * ldb + j)), x);
        }
#pragma unroll
        for (int j = 0; j < NN; ++j)
            if (j < N) st_elem<F>(C, row * ldc + j, f.acc_reduce(acc[j]));
    }
}

// C (M x N) = A (M x K) @ B (K x N), M <= SKINNY_MAX: a pack of columns per thread, K split over blockIdx.y;
// partial[(ks * M + m) * N + j], summed by k_vecmat_final
template <class F, int MM, bool VEC, int UNR = 4>
__global__ __launch_bounds__(BLOCK) void k_vecmat_partial(F f, const typename F::elem* __restrict__ A, size_t lda,
                                                           const typename F::elem* __restrict__ B, size_t ldb,
                                                           typename F::word* __restrict__ partial, int M, int K, int N,
                                                           int kchunk) {
    typedef Pack<typename F::word> P;
    typedef typename MemPack<F>::type MP;
    typedef typename F::word W;
    constexpr int CW = VEC ? P::N : 1;                              // columns per thread
    const int j = (blockIdx.x * BLOCK + threadIdx.x) * CW;
    if (j >= N) return;
    const int k0 = blockIdx.y * kchunk;
    const int k1 = k0 + kchunk < K ? k0 + kchunk : K;
    typename F::acc acc[MM][CW];
    W total[MM][CW];
    bool have = false;
    int cnt = 0;
#pragma unroll
    for (int mi = 0; mi < MM; ++mi)
#pragma unroll
        for (int q = 0; q < CW; ++q) f.acc_zero(acc[mi][q]);
    auto flush = [&]() {
#pragma unroll
        for (int mi = 0; mi < MM; ++mi)
#pragma unroll
            for (int q = 0; q < CW; ++q) {
                W part = f.acc_reduce(acc[mi][q]);
                total[mi][q] = have ? f.add(total[mi][q], part) : part;
                f.acc_zero(acc[mi][q]);
            }
        have = true;
        cnt = 0;
    };
    auto load_b = [&](int kk, W (&b)[CW]) {
        if constexpr (VEC) {
            const P bp = ldg<true>(reinterpret_cast<const MP*>(B + (size_t)kk * ldb + j));     // coalesced across the block
#pragma unroll
            for (int q = 0; q < CW; ++q) b[q] = bp.w[q];
        } else {
            b[0] = ld_elem<F>(B, (size_t)kk * ldb + j);
        }
    };
    auto macs = [&](int kk, const W (&b)[CW]) {
#pragma unroll
        for (int mi = 0; mi < MM; ++mi)
            if (mi < M) {
                const W ap = f.prep(ld_elem<F>(A, (size_t)mi * lda + kk));                      // wave-uniform operand
#pragma unroll
                for (int q = 0; q < CW; ++q) f.acc_mac(acc[mi][q], ap, b[q]);
            }
    };
    int kk = k0;
    for (; kk + UNR <= k1; kk += UNR) {              // UNR rows of B in flight per thread
        W b[UNR][CW];
#pragma unroll
        for (int u = 0; u < UNR; ++u) load_b(kk + u, b[u]);
#pragma unroll
        for (int u = 0; u < UNR; ++u) macs(kk + u, b[u]);
        cnt += UNR;
        if (cnt >= SKINNY_FLUSH) flush();
    }
    for (; kk < k1; ++kk) {
        W b0[CW];
        load_b(kk, b0);
        macs(kk, b0);
        if (++cnt >= SKINNY_FLUSH) flush();
    }
    flush();
#pragma unroll
    for (int mi = 0; mi < MM; ++mi)
        if (mi < M)
#pragma unroll
            for (int q = 0; q < CW; ++q)
                if (j + q < N) partial[((size_t)blockIdx.y * M + mi) * N + j + q] = total[mi][q];
}

template <class F>
__global__ __launch_bounds__(BLOCK) void k_vecmat_final(F f, const typename F::word* __restrict__ partial, int KS, int M,
                                                         int N, typename F::elem* __restrict__ C, size_t ldc) {
    typedef typename F::word W;
    const size_t idx = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    if (idx >= (size_t)M * N) return;
    const int mi = (int)(idx / N), j = (int)(idx % N);
    // KS can be ~128: eight independent chains so that the loads overlap instead of forming one dependent sequence
    W r[8];
    const size_t mn = (size_t)M * N;
    int s = 0;
    if (KS >= 8) {
#pragma unroll
        for (int u = 0; u < 8; ++u) r[u] = partial[(size_t)u * mn + idx];
        for (s = 8; s + 8 <= KS; s += 8) {
            W t[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) t[u] = partial[(size_t)(s + u) * mn + idx];
#pragma unroll
            for (int u = 0; u < 8; ++u) r[u] = f.add(r[u], t[u]);
        }
        r[0] = f.add(f.add(f.add(r[0], r[1]), f.add(r[2], r[3])), f.add(f.add(r[4], r[5]), f.add(r[6], r[7])));
    } else {
        r[0] = partial[idx];
        s = 1;
    }
    for (; s < KS; ++s) r[0] = f.add(r[0], partial[(size_t)s * mn + idx]);
    st_elem<F>(C, (size_t)mi * ldc + j, r[0]);
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

// ---- launch plumbing -------------------------------------------------------
struct LaunchCfg {
    int blocks_per_cu;  // 0 = uncapped grid: one 16-byte pack per thread (default, measured best)
    int num_cu;
    int nt;             // non-temporal loads/stores (default 1)
};
LaunchCfg launch_cfg(int device);

inline unsigned grid_for(size_t iters, const LaunchCfg& lc) {
    size_t want = (iters + BLOCK - 1) / BLOCK;
    size_t cap = lc.blocks_per_cu > 0 ? (size_t)lc.blocks_per_cu * (size_t)lc.num_cu : (size_t)0x7fffffff;
    if (want < 1) want = 1;
    return (unsigned)(want < cap ? want : cap);
}

inline bool aligned16(const void* p) { return ((uintptr_t)p & 15u) == 0; }

// host-side table of launchers for one policy type; the context stores the
// policy blob and a pointer to this table.
struct FieldOps {
    int (*ew2)(const void* F, int device, int op, const void* a, const void* b, void* o, size_t n,
               hipStream_t st);
    int (*ew1)(const void* F, int device, int op, const void* a, const uint64_t* scalar2, void* o,
               size_t n, hipStream_t st);
    int (*muladd)(const void* F, int device, const void* a, const void* b, const void* c, void* o,
                  size_t n, hipStream_t st);
    // coef == nullptr && rng != nullptr: coefficients are drawn in-kernel from the keystream
    int (*split)(const void* F, int device, const void* a, const void* b, const void* coef,
                 size_t cstride, int t, int m, void* out, size_t ostride, size_t n, hipStream_t st,
                 const RngArgs* rng);
    int (*rng_coeffs)(const void* F, int device, void* coef, size_t cstride, int t, size_t n, hipStream_t st,
                      const RngArgs* rng);
    int (*recombine)(const void* F, int device, const void* const* rows, const uint64_t* lam2, int k,
                     int w, void* out, size_t ostride, size_t n, hipStream_t st);
    int (*pow)(const void* F, int device, const void* a, const ExpArgs* ex, void* out, size_t n, hipStream_t st);
    int (*inv)(const void* F, int device, const void* a, const ExpArgs* ex, void* out, size_t n, int* flag,
               hipStream_t st);
    int (*matmul)(const void* F, int device, const void* A, size_t lda, const void* B, size_t ldb, void* C,
                  size_t ldc, int M, int K, int N, void* workspace, size_t workspace_bytes, int mod_bits,
                  hipStream_t st);
    int (*dot)(const void* F, int device, const void* a, const void* b, void* out, void* workspace, size_t n,
               hipStream_t st);
    // nbatch > 1: gridDim.y independent gates in one launch, operands / outputs of gate y at element offsets y*yA, y*yB, y*yO
    int (*gate)(const void* F, int device, const void* const* rowsA, const uint64_t* lamA2, int kA,
                const void* const* rowsB, const uint64_t* lamB2, int kB, int t, int m, void* out, size_t ostride,
                size_t n, hipStream_t st, const RngArgs* rng, int nbatch, size_t yA, size_t yB, size_t yO);
    int (*sqrt_cl)(const void* F, int device, const void* a, const ExpArgs* eleg, const ExpArgs* elad, void* out, size_t n,
                   hipStream_t st);
    int (*gauss)(const void* F, int device, void* A, int n, int ncols, size_t batch, int det_mode, const ExpArgs* ex,
                 void* det, int* sing, hipStream_t st);
    int (*group_matvec)(const void* F, int device, const uint64_t* m2, const uint64_t* bias2, int r, int g,
                        const void* in, void* out, size_t ngroups, hipStream_t st);
    int (*beaver)(const void* F, int device, const void* z, const void* x, const void* y, const void* d, const void* e,
                  void* out, int add_de, size_t n, hipStream_t st);
    int (*prss)(const void* F, int device, const void* const* streams, int ks, int d, int l, int mask_bits,
                const uint64_t* weights2, const uint64_t* r2, int accumulate, void* out, size_t n, hipStream_t st);
    // keys40: ks x (32-byte ChaCha key + 8-byte nonce)
    int (*prss_chacha)(const void* F, int device, const uint8_t* keys40, int ks, int d, int l, int mask_bits, int rounds,
                       const uint64_t* weights2, const uint64_t* r2, int accumulate, void* out, size_t n, hipStream_t st);
};

// Host scalars (Lagrange coefficients, constants, matrix entries) cross the C ABI as little-endian 64-bit limbs:
// 2 per scalar, 3 for the three-limb prime fields (ffgpu_ctx_scalar_limbs).
template <class F>
constexpr int scalar_limbs() {
    return sizeof(typename F::word) == 24 ? 3 : 2;
}
// scalar number idx of a host array -> policy word (broadcast for packed fields)
template <class F>
inline typename F::word word_at(const F& f, const uint64_t* base, size_t idx);
template <class F>
inline typename F::word word_from_limbs(const F& f, uint64_t lo, uint64_t hi) {
    if constexpr (sizeof(typename F::word) == 24) {
        typename F::word w;
        w.lo = lo;
        w.mid = hi;
        w.hi = 0;
        return w;
    } else if constexpr (sizeof(typename F::word) == 16) {
        typename F::word w;
        w.lo = lo;
        w.hi = hi;
        return w;
    } else if constexpr (F::EPW == 4) {
        uint32_t b = (uint32_t)(lo & 0xffu);
        return b * 0x01010101u;
    } else {
        return (typename F::word)lo;
    }
}
template <class F>
inline typename F::word word_at(const F& f, const uint64_t* base, size_t idx) {
    constexpr int SL = scalar_limbs<F>();
    const uint64_t* l = base + idx * SL;
    if constexpr (SL == 3) {
        typename F::word w;
        w.lo = l[0];
        w.mid = l[1];
        w.hi = l[2];
        return w;
    } else {
        return word_from_limbs<F>(f, l[0], l[1]);
    }
}

template <class F>
inline typename F::word prep_const(const F& f, typename F::word c) {
    return f.prep(c);
}

#define FFGPU_CHECK_LAUNCH()                      \
    do {                                          \
        hipError_t e__ = hipGetLastError();       \
        if (e__ != hipSuccess) return (int)e__ | 0x10000; \
    } while (0)

template <class F>
struct Launchers {
    typedef typename F::elem E;
    typedef typename F::word W;
    enum { EPV = Pack<W>::N * F::EPW };  // elements per pack (one lane's access)
    // dwordx3 needs dword alignment only; three-limb elements go as three dwordx2
    enum { PACK_ALIGN = sizeof(E) == 12 ? 4 : sizeof(E) == 24 ? 8 : 16 };
    static bool al(const void* p) { return ((uintptr_t)p & (PACK_ALIGN - 1)) == 0; }
    // (a member function, not a lambda inside `matmul`: clang does not emit the host stub of a kernel specialisation
    // that is only named inside a generic lambda's discarded-branch neighbourhood)
    template <bool BRAW>
    static void launch_glds(const F& f, dim3 grid, hipStream_t st, const int8_t* Ap, const int8_t* Bp, E* out, size_t out_ld, int M, int N,
                            int Kp, int kb, int ke, int acc_, int kslice, size_t zs, const E* Braw, size_t ldb, uint64_t pmod) {
        if constexpr (F::EPW == 1 && !F::BINARY && sizeof(E) == 8) {
            const size_t lds = (size_t)(BRAW ? 8 : 6) * GLDS_TILE;
            static bool attr_done = false;
            if (!attr_done) {
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_limb_gemm_glds<F, BRAW>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                attr_done = true;
            }
            hipLaunchKernelGGL((k_limb_gemm_glds<F, BRAW>), grid, dim3(BLOCK), lds, st, f, Ap, Bp, out, out_ld, M, N, Kp, kb, ke, acc_,
                               kslice, zs, Braw, ldb, pmod);
        }
    }
    static bool limb_glds() {                       // FFGPU_MM_GLDS=0: the register-staged product kernel (read per call)
        const char* e = getenv("FFGPU_MM_GLDS");
        return !(e && atoi(e) == 0);
    }
    static bool limb_braw() {                       // FFGPU_MM_BRAW=0: always go through digit planes of B (A/B measurements)
        static int v = -1;
        if (v < 0) {
            const char* e = getenv("FFGPU_MM_BRAW");
            v = e ? atoi(e) : 1;
        }
        return v != 0;
    }
    static bool stride_ok(size_t stride) { return (stride * sizeof(E)) % PACK_ALIGN == 0; }

    template <int OP>
    static void go_ew2(const F& f, const LaunchCfg& lc, const E* a, const E* b, E* o, size_t n, hipStream_t st) {
        bool vec = al(a) && al(b) && al(o);
        size_t nvec = vec ? n / EPV : 0;
        unsigned grid = grid_for(nvec ? nvec : n, lc);
        if (lc.nt)
            hipLaunchKernelGGL((k_ew2<F, OP, true>), dim3(grid), dim3(BLOCK), 0, st, f, a, b, o, nvec, n);
        else
            hipLaunchKernelGGL((k_ew2<F, OP, false>), dim3(grid), dim3(BLOCK), 0, st, f, a, b, o, nvec, n);
    }
    static int ew2(const void* Fp, int device, int op, const void* a, const void* b, void* o, size_t n,
                   hipStream_t st) {
        const F& f = *reinterpret_cast<const F*>(Fp);
        LaunchCfg lc = launch_cfg(device);
        const E* A = (const E*)a;
        const E* B = (const E*)b;
        E* O = (E*)o;
        switch (op) {
            case OP_ADD: go_ew2<OP_ADD>(f, lc, A, B, O, n, st); break;
            case OP_SUB: go_ew2<OP_SUB>(f, lc, A, B, O, n, st); break;
            case OP_MUL: go_ew2<OP_MUL>(f, lc, A, B, O, n, st); break;
            default: return 1;
        }
        FFGPU_CHECK_LAUNCH();
        return 0;
    }

    template <int OP>
    static void go_ew1(const F& f, const LaunchCfg& lc, const E* a, W s, E* o, size_t n, hipStream_t st) {
        bool vec = al(a) && al(o);
        size_t nvec = vec ? n / EPV : 0;
        unsigned grid = grid_for(nvec ? nvec : n, lc);
        if (lc.nt)
            hipLaunchKernelGGL((k_ew1<F, OP, true>), dim3(grid), dim3(BLOCK), 0, st, f, a, s, o, nvec, n);
        else
            hipLaunchKernelGGL((k_ew1<F, OP, false>), dim3(grid), dim3(BLOCK), 0, st, f, a, s, o, nvec, n);
    }
    static int ew1(const void* Fp, int device, int op, const void* a, const uint64_t* scalar2, void* o,
                   size_t n, hipStream_t st) {
        const F& f = *reinterpret_cast<const F*>(Fp);
        LaunchCfg lc = launch_cfg(device);
        W s = scalar2 ? word_at<F>(f, scalar2, 0) : word_from_limbs<F>(f, 0, 0);
        const E* A = (const E*)a;
        E* O = (E*)o;
        switch (op) {
            case OP_ADD: go_ew1<OP_ADD>(f, lc, A, s, O, n, st); break;
            case OP_RSUB: go_ew1<OP_RSUB>(f, lc, A, s, O, n, st); break;
            case OP_MUL: go_ew1<OP_MUL>(f, lc, A, s, O, n, st); break;
            case OP_NEG: go_ew1<OP_NEG>(f, lc, A, s, O, n, st); break;
            case OP_REDUCE: go_ew1<OP_REDUCE>(f, lc, A, s, O, n, st); break;
            case OP_COPY: go_ew1<OP_COPY>(f, lc, A, s, O, n, st); break;
            default: return 1;
        }
        FFGPU_CHECK_LAUNCH();
        return 0;
    }

    static int muladd(const void* Fp, int device, const void* a, const void* b, const void* c, void* o,
                      size_t n, hipStream_t st) {
        const F& f = *reinterpret_cast<const F*>(Fp);
        LaunchCfg lc = launch_cfg(device);
        bool vec = al(a) && al(b) && al(c) && al(o);
        size_t nvec = vec ? n / EPV : 0;
        unsigned grid = grid_for(nvec ? nvec : n, lc);
        if (lc.nt)
            hipLaunchKernelGGL((k_muladd<F, true>), dim3(grid), dim3(BLOCK), 0, st, f, (const E*)a,
                               (const E*)b, (const E*)c, (E*)o, nvec, n);
        else
            hipLaunchKernelGGL((k_muladd<F, false>), dim3(grid), dim3(BLOCK), 0, st, f, (const E*)a,
                               (const E*)b, (const E*)c, (E*)o, nvec, n);
        FFGPU_CHECK_LAUNCH();
        return 0;
    }

    template <int T, bool FUSE, bool RNG, bool REC = false>
    static void go_split(const F& f, unsigned grid, bool nt, const E* a, const E* b, const E* coef,
                         size_t cstride, int m, E* out, size_t ostride, size_t nvec, size_t n, hipStream_t st,
                         const RngArgs& ra, const GateSrc<F>& gs, unsigned gy = 1) {
        if (nt || RNG)
            hipLaunchKernelGGL((k_split<F, T, FUSE, true, RNG, REC>), dim3(grid, gy), dim3(BLOCK), 0, st, f, a, b,
                               coef, cstride, m, out, ostride, nvec, n, ra, gs);
        else
            hipLaunchKernelGGL((k_split<F, T, FUSE, false, false>), dim3(grid), dim3(BLOCK), 0, st, f, a, b,
                               coef, cstride, m, out, ostride, nvec, n, ra, gs);
    }
    template <bool FUSE, bool RNG>
    static int split_t(const F& f, const LaunchCfg& lc, const E* a, const E* b, const E* coef, size_t cstride,
                       int t, int m, E* out, size_t ostride, size_t n, hipStream_t st, const RngArgs& ra_in) {
        if (t > MAXT) {
            RngArgs ra = ra_in;
            unsigned grid = grid_for(n, lc);
            ra.release = !ra.no_advance && grid <= RNG_RELEASE_MAX_GRID;
            hipLaunchKernelGGL((k_split_any<F, FUSE, RNG>), dim3(grid), dim3(BLOCK), 0, st, f, a, b, coef, cstride,
                               t, m, out, ostride, n, ra);
            if (RNG && ra.dev_key && !ra.release && !ra.no_advance)
                hipLaunchKernelGGL((k_rng_advance<0>), dim3(1), dim3(1), 0, st, const_cast<RngKey*>(ra.dev_key), 1u);
            return 0;
        }
        bool vec = al(a) && (!FUSE || al(b)) && al(out) &&
                   (stride_ok(ostride) || m <= 1) &&
                   (RNG || t == 0 || (al(coef) && (stride_ok(cstride) || t <= 1)));
        size_t nvec = vec ? n / EPV : 0;
        // RNG kernels serve up to 4 packs per thread (RngLayout::G); a slightly larger grid is harmless.
        // Below ~2.6e5 packs the grouped loop cannot fill the chip: one pack per thread instead (ra.spread).
        RngArgs ra = ra_in;
        const bool spread = RNG && nvec > 0 && nvec < 262144;
        ra.spread = spread ? 1 : 0;
        unsigned grid = grid_for(nvec ? (RNG && !spread ? (n / EPV + 2) / 2 : nvec) : n, lc);
        ra.release = !ra.no_advance && grid <= RNG_RELEASE_MAX_GRID;
        bool nt = lc.nt != 0;
        GateSrc<F> gs;
        memset(&gs, 0, sizeof(gs));
        switch (t) {
            case 0: go_split<0, FUSE, false>(f, grid, nt, a, b, coef, cstride, m, out, ostride, nvec, n, st, ra, gs); break;
            case 1: go_split<1, FUSE, RNG>(f, grid, nt, a, b, coef, cstride, m, out, ostride, nvec, n, st, ra, gs); break;
            case 2: go_split<2, FUSE, RNG>(f, grid, nt, a, b, coef, cstride, m, out, ostride, nvec, n, st, ra, gs); break;
            case 3: go_split<3, FUSE, RNG>(f, grid, nt, a, b, coef, cstride, m, out, ostride, nvec, n, st, ra, gs); break;
            case 4: go_split<4, FUSE, RNG>(f, grid, nt, a, b, coef, cstride, m, out, ostride, nvec, n, st, ra, gs); break;
            default: return 1;
        }
        if (RNG && t > 0 && ra.dev_key && !ra.release && !ra.no_advance)
            hipLaunchKernelGGL((k_rng_advance<0>), dim3(1), dim3(1), 0, st, const_cast<RngKey*>(ra.dev_key), 1u);
        return 0;
    }
    // fused chain gate: both factors given as recombinations (GateSrc), product re-shared with the device CSPRNG
    static int gate(const void* Fp, int device, const void* const* rowsA, const uint64_t* lamA2, int kA,
                    const void* const* rowsB, const uint64_t* lamB2, int kB, int t, int m, void* out, size_t ostride,
                    size_t n, hipStream_t st, const RngArgs* rng, int nbatch, size_t yA, size_t yB, size_t yO) {
        const F& f = *reinterpret_cast<const F*>(Fp);
        if (t < 1 || t > 3 || kA < 1 || kA > GATE_MAXK || kB < 0 || kB > GATE_MAXK || !rng) return 2;
        if (nbatch < 1 || nbatch > 255) return 2;
        LaunchCfg lc = launch_cfg(device);
        GateSrc<F> gs;
        memset(&gs, 0, sizeof(gs));
        bool vec = al(out) && (stride_ok(ostride) || m <= 1);
        if (nbatch > 1) {
            gs.yA = yA;
            gs.yB = yB;
            gs.yO = yO;
            vec = vec && stride_ok(yA) && stride_ok(yO) && (kB == 0 || stride_ok(yB));
        }
        for (int j = 0; j < kA; ++j) {
            gs.rowsA[j] = (const E*)rowsA[j];
            gs.lamA[j] = f.prep(word_at<F>(f, lamA2, j));
            vec = vec && al(rowsA[j]);
        }
        for (int j = 0; j < kB; ++j) {
            gs.rowsB[j] = (const E*)rowsB[j];
            gs.lamB[j] = f.prep(word_at<F>(f, lamB2, j));
            vec = vec && al(rowsB[j]);
        }
        gs.kA = kA;
        gs.kB = kB;
        gs.square = kB == 0;
        constexpr int SLG = scalar_limbs<F>();
        gs.plainA = kA == 1 && lamA2[0] == 1 && lamA2[1] == 0 && (SLG < 3 || lamA2[SLG - 1] == 0);
        gs.plainB = kB == 1 && lamB2[0] == 1 && lamB2[1] == 0 && (SLG < 3 || lamB2[SLG - 1] == 0);
        size_t nvec = vec ? n / EPV : 0;
        RngArgs ra = *rng;
        const bool spread = nvec > 0 && nvec < 262144;
        ra.spread = spread ? 1 : 0;
        unsigned grid = grid_for(nvec ? (!spread ? (n / EPV + 2) / 2 : nvec) : n, lc);
        const unsigned gy = (unsigned)nbatch;
        ra.release = !ra.no_advance && (size_t)grid * gy <= RNG_RELEASE_MAX_GRID;
        E* o = (E*)out;
        switch (t) {
            case 1: go_split<1, true, true, true>(f, grid, true, nullptr, nullptr, nullptr, 0, m, o, ostride, nvec, n, st, ra, gs, gy); break;
            case 2: go_split<2, true, true, true>(f, grid, true, nullptr, nullptr, nullptr, 0, m, o, ostride, nvec, n, st, ra, gs, gy); break;
            default: go_split<3, true, true, true>(f, grid, true, nullptr, nullptr, nullptr, 0, m, o, ostride, nvec, n, st, ra, gs, gy); break;
        }
        if (ra.dev_key && !ra.release && !ra.no_advance)
            hipLaunchKernelGGL((k_rng_advance<0>), dim3(1), dim3(1), 0, st, const_cast<RngKey*>(ra.dev_key), 1u);
        FFGPU_CHECK_LAUNCH();
        return 0;
    }
    static int split(const void* Fp, int device, const void* a, const void* b, const void* coef,
                     size_t cstride, int t, int m, void* out, size_t ostride, size_t n, hipStream_t st,
                     const RngArgs* rng) {
        const F& f = *reinterpret_cast<const F*>(Fp);
        LaunchCfg lc = launch_cfg(device);
        RngArgs ra;
        memset(&ra, 0, sizeof(ra));
        int rc;
        if (rng) {
            ra = *rng;
            rc = b ? split_t<true, true>(f, lc, (const E*)a, (const E*)b, nullptr, 0, t, m, (E*)out, ostride, n, st, ra)
                   : split_t<false, true>(f, lc, (const E*)a, nullptr, nullptr, 0, t, m, (E*)out, ostride, n, st, ra);
        } else {
            rc = b ? split_t<true, false>(f, lc, (const E*)a, (const E*)b, (const E*)coef, cstride, t, m, (E*)out,
                                          ostride, n, st, ra)
                   : split_t<false, false>(f, lc, (const E*)a, nullptr, (const E*)coef, cstride, t, m, (E*)out,
                                           ostride, n, st, ra);
        }
        if (rc) return rc;
        FFGPU_CHECK_LAUNCH();
        return 0;
    }
    static int rng_coeffs(const void* Fp, int device, void* coef, size_t cstride, int t, size_t n, hipStream_t st,
                          const RngArgs* rng) {
        const F& f = *reinterpret_cast<const F*>(Fp);
        LaunchCfg lc = launch_cfg(device);
        E* C = (E*)coef;
        size_t npacks = (n + EPV - 1) / EPV;
        unsigned grid = grid_for(npacks, lc);
        if (t > MAXT) {
            hipLaunchKernelGGL((k_rng_coeffs_any<F>), dim3(grid), dim3(BLOCK), 0, st, f, C, cstride, t, n, *rng);
            FFGPU_CHECK_LAUNCH();
            return 0;
        }
        bool vec = al(coef) && (stride_ok(cstride) || t <= 1);
        size_t nvec = vec ? n / EPV : 0;
        switch (t) {
            case 1: hipLaunchKernelGGL((k_rng_coeffs<F, 1>), dim3(grid), dim3(BLOCK), 0, st, f, C, cstride, nvec, n, *rng); break;
            case 2: hipLaunchKernelGGL((k_rng_coeffs<F, 2>), dim3(grid), dim3(BLOCK), 0, st, f, C, cstride, nvec, n, *rng); break;
            case 3: hipLaunchKernelGGL((k_rng_coeffs<F, 3>), dim3(grid), dim3(BLOCK), 0, st, f, C, cstride, nvec, n, *rng); break;
            case 4: hipLaunchKernelGGL((k_rng_coeffs<F, 4>), dim3(grid), dim3(BLOCK), 0, st, f, C, cstride, nvec, n, *rng); break;
            default: return 1;
        }
        FFGPU_CHECK_LAUNCH();
        return 0;
    }

    template <int K>
    static void go_rec(const F& f, const LaunchCfg& lc, const void* const* rows, const uint64_t* lam2, int w,
                       E* out, size_t ostride, size_t n, hipStream_t st) {
        RecArgs<F, K> ra;
        bool vec = al(out) && (stride_ok(ostride) || w <= 1);
        for (int j = 0; j < K; ++j) {
            ra.rows[j] = (const E*)rows[j];
            vec = vec && al(rows[j]);
        }
        for (int r = 0; r < w; ++r)
            for (int j = 0; j < K; ++j) {
                ra.lam[r * K + j] = f.prep(word_at<F>(f, lam2, (size_t)r * K + j));
            }
        for (int i = w * K; i < MAXW * K; ++i) ra.lam[i] = ra.lam[0];
        size_t nvec = vec ? n / EPV : 0;
        unsigned grid = grid_for(nvec ? nvec : n, lc);
        if (lc.nt)
            hipLaunchKernelGGL((k_recombine<F, K, true>), dim3(grid), dim3(BLOCK), 0, st, f, ra, w, out, ostride,
                               nvec, n);
        else
            hipLaunchKernelGGL((k_recombine<F, K, false>), dim3(grid), dim3(BLOCK), 0, st, f, ra, w, out, ostride,
                               nvec, n);
    }
    static int recombine(const void* Fp, int device, const void* const* rows, const uint64_t* lam2, int k,
                         int w, void* out, size_t ostride, size_t n, hipStream_t st) {
        const F& f = *reinterpret_cast<const F*>(Fp);
        LaunchCfg lc = launch_cfg(device);
        E* O = (E*)out;
        if (k > MAXK) {
            if (k > MAXK_ANY) return 2;
            for (int r = 0; r < w; ++r) {
                RecArgsAny<F> ra;
                for (int j = 0; j < k; ++j) {
                    ra.rows[j] = (const E*)rows[j];
                    ra.lam[j] = f.prep(word_at<F>(f, lam2, (size_t)r * k + j));
                }
                for (int j = k; j < MAXK_ANY; ++j) {
                    ra.rows[j] = ra.rows[0];
                    ra.lam[j] = ra.lam[0];
                }
                unsigned grid = grid_for(n, lc);
                hipLaunchKernelGGL((k_recombine_any<F>), dim3(grid), dim3(BLOCK), 0, st, f, ra, k,
                                   O + (size_t)r * ostride, n);
            }
            FFGPU_CHECK_LAUNCH();
            return 0;
        }
        for (int r0 = 0; r0 < w; r0 += MAXW) {
            int wc = (w - r0) < MAXW ? (w - r0) : MAXW;
            const uint64_t* l = lam2 + scalar_limbs<F>() * (size_t)r0 * k;
            E* o = O + (size_t)r0 * ostride;
            switch (k) {
                case 1: go_rec<1>(f, lc, rows, l, wc, o, ostride, n, st); break;
                case 2: go_rec<2>(f, lc, rows, l, wc, o, ostride, n, st); break;
                case 3: go_rec<3>(f, lc, rows, l, wc, o, ostride, n, st); break;
                case 4: go_rec<4>(f, lc, rows, l, wc, o, ostride, n, st); break;
                case 5: go_rec<5>(f, lc, rows, l, wc, o, ostride, n, st); break;
                case 6: go_rec<6>(f, lc, rows, l, wc, o, ostride, n, st); break;
                case 7: go_rec<7>(f, lc, rows, l, wc, o, ostride, n, st); break;
                case 8: go_rec<8>(f, lc, rows, l, wc, o, ostride, n, st); break;
                case 9: go_rec<9>(f, lc, rows, l, wc, o, ostride, n, st); break;
                default: return 1;
            }
        }
        FFGPU_CHECK_LAUNCH();
        return 0;
    }

    static int pow(const void* Fp, int device, const void* a, const ExpArgs* ex, void* out, size_t n,
                   hipStream_t st) {
        const F& f = *reinterpret_cast<const F*>(Fp);
        LaunchCfg lc = launch_cfg(device);
        bool vec = al(a) && al(out);
        size_t nvec = vec ? n / EPV : 0;
        unsigned grid = grid_for(nvec ? nvec : n, lc);
        hipLaunchKernelGGL((k_pow<F, true>), dim3(grid), dim3(BLOCK), 0, st, f, (const E*)a, *ex, (E*)out, nvec, n);
        FFGPU_CHECK_LAUNCH();
        return 0;
    }
    static int inv(const void* Fp, int device, const void* a, const ExpArgs* ex, void* out, size_t n, int* flag,
                   hipStream_t st) {
        const F& f = *reinterpret_cast<const F*>(Fp);
        LaunchCfg lc = launch_cfg(device);
        bool vec = al(a) && al(out);
        size_t nvec = vec ? n / EPV : 0;
        // packs per thread: ONE exponentiation (70 products for 2^61 - 1) is shared by G x CH packs, and the
        // G x CH x N prefix words stay in registers (two waves per SIMD at CH = 8..12 for one-word fields)
        if constexpr (F::EPW == 1 && sizeof(W) == 8) {
            // One exponentiation (70 products for 2^61 - 1) is shared by the CH x G packs of a thread.  Round 4: k_inv_fast
            // (full batches without predicates, second reads in a window, exponentiation without its window table when the
            // exponent allows it) at 8 x 2 packs = 32 elements per thread and three waves per SIMD: 41.6 us against 51.6 us
            // for the round-3 kernel at n = 10^7 over 2^61 - 1 (profiles/r04_alu.md: thirteen shapes measured; more packs
            // per thread spill or fall to two waves, fewer pay more exponentiations).  FFGPU_INV_VARIANT=0 runs the round-3
            // kernel (A/B measurements).
            const char* e = getenv("FFGPU_INV_VARIANT");
            const int variant = e ? atoi(e) : 1;
            // (prime fields only: the GF(2^n) product has run-time loops, its prefix array lives in scratch memory either way,
            // and the round-3 kernel needs less of it -- 272-336 B against 528-624 B per thread)
            if (variant > 0 && !F::BINARY && nvec >= (size_t)BLOCK * 64) {
                if (pow_lean_ok(*ex)) return launch_inv_fast<8, 2, 6, 3, true>(f, a, ex, out, nvec, n, flag, st);
                return launch_inv_fast<8, 2, 3, 1, false>(f, a, ex, out, nvec, n, flag, st);
            }
            // All waves of the launch take the same time and two fit on a SIMD, so the launch runs in ROUNDS of
            // 2 x 4 x num_cu waves: 10^7 elements at CH = 8 are 4883 waves = 2.4 rounds -- three rounds of time for
            // 2.4 of work (measured: 56 us).  More packs per thread amortise the exponentiation better AND change the
            // number of rounds; pick the CH with the least rounds x (products per thread).
            const size_t slots = (size_t)lc.num_cu * 4 * 2;
            int best = 8;
            double best_cost = 0;
            for (int ch : {8, 10}) {                    // (CH = 12: 296 VGPRs, one wave per SIMD)
                const size_t waves = (nvec / (size_t)(ch * 2) + 63) / 64 + 1;
                const size_t rounds = (waves + slots - 1) / slots;
                const double cost = (double)rounds * (3.0 * ch * 2 * (double)EPV + 73.0);
                if (ch == 8 || cost < best_cost * 0.97) {
                    best = ch;
                    best_cost = ch == 8 ? cost : (cost < best_cost ? cost : best_cost);
                }
            }
            if (best == 10) return launch_inv<10, 2>(f, lc, a, ex, out, nvec, n, flag, st);
            return launch_inv<8, 2>(f, lc, a, ex, out, nvec, n, flag, st);
        } else {
            constexpr int CH = F::EPW > 1 ? 2 : 8;          // packed bytes: 8 words per batch (zero mask)
            return launch_inv<CH, 1>(f, lc, a, ex, out, nvec, n, flag, st);
        }
    }
    // square-and-multiply after the leading run costs popcount(tail) products, the window table 8 up front and one per
    // window: lean when the tail holds few set bits (q - 2 of every 2^k - c prime: a run of ones and a short tail)
    static bool pow_lean_ok(const ExpArgs& ex) {
        int i = ex.nbits - 1;
        auto bit = [&](int b) { return (int)((ex.e[b >> 6] >> (b & 63)) & 1u); };
        while (i >= 0 && bit(i)) --i;                 // the leading run
        if (ex.nbits - 1 - i < 12) i = ex.nbits - 2;  // (short runs are not raised by doubling: everything is tail)
        int ones = 0;
        for (int b = i; b >= 0; --b) ones += bit(b);
        return ones <= 6;
    }
    template <int CH, int G, int WIN, int WAVES, bool LEAN, int WIN1 = 0>
    static int launch_inv_fast(const F& f, const void* a, const ExpArgs* ex, void* out, size_t nvec, size_t n, int* flag,
                               hipStream_t st) {
        if constexpr (F::EPW == 1 && sizeof(W) == 8) {
            const size_t per_block = (size_t)BLOCK * CH * G;
            const size_t nfull = nvec / per_block;
            const size_t rest = nvec % per_block;
            const unsigned grid = (unsigned)nfull + (unsigned)((rest + BLOCK - 1) / BLOCK) + ((rest == 0 && n > nvec * EPV) ? 1u : 0u);
            hipLaunchKernelGGL((k_inv_fast<F, CH, G, WIN, WAVES, LEAN, WIN1>), dim3(grid), dim3(BLOCK), 0, st, f, (const E*)a, *ex, (E*)out,
                               nvec, n, (unsigned)nfull, flag);
            FFGPU_CHECK_LAUNCH();
        }
        return 0;
    }
    template <int CH, int G>
    static int launch_inv(const F& f, const LaunchCfg& lc, const void* a, const ExpArgs* ex, void* out, size_t nvec, size_t n,
                          int* flag, hipStream_t st) {
        size_t iters = nvec ? (nvec + CH * G - 1) / (CH * G) : n;
        unsigned grid = grid_for(iters, lc);
        hipLaunchKernelGGL((k_inv_batch<F, CH, G, true>), dim3(grid), dim3(BLOCK), 0, st, f, (const E*)a, *ex, (E*)out,
                           nvec, n, flag);
        FFGPU_CHECK_LAUNCH();
        return 0;
    }

    // FFGPU_SKINNY_V2=0 selects the first-generation skinny kernels (A/B measurements)
    static bool skinny_v2() {
        static int v = -1;
        if (v < 0) {
            const char* e = getenv("FFGPU_SKINNY_V2");
            v = e ? atoi(e) : 1;
        }
        return v != 0;
    }
    // skinny shapes (one output dimension <= 8): HBM-bound kernels that read the big operand once
    template <int NN>
    static void go_matvec(const F& f, const E* A, size_t lda, const E* B, size_t ldb, E* C, size_t ldc, int M, int K,
                          int N, hipStream_t st) {
        if (K <= 32 && M >= 1024) {               // short rows: one thread per row
            unsigned grid = (unsigned)(((size_t)M + BLOCK - 1) / BLOCK);
            hipLaunchKernelGGL((k_matvec_short_rows<F, NN>), dim3(grid), dim3(BLOCK), 0, st, f, A, lda, B, ldb, C, ldc, M, K, N);
            return;
        }
        const int vec = al(A) && stride_ok(lda);
        const int bvec = al(B) && stride_ok(ldb) && (N % (int)Pack<W>::N == 0) && sizeof(E) != 12;
        if constexpr (NN <= 2 && sizeof(E) != 12) {
            // long rows, one or two columns: R = 2 rows per workgroup share every load of B (measured at 4096^2 / 8192^2:
            // R = 1 32.5 / 117 us, R = 2 27.4 / 85 us, R = 4 28.3 / 99 us, R = 8 35.8 / 112 us)
            constexpr int R = 2;
            if (skinny_v2() && vec && K >= 1024 && M >= 1024 * R) {
                const int bpack = (N == 1 && ldb == 1 && al(B)) ? 1 : 0;
                hipLaunchKernelGGL((k_matvec_rows_r<F, NN, R>), dim3((unsigned)((M + R - 1) / R)), dim3(BLOCK), 0, st, f, A, lda, B,
                                   ldb, C, ldc, M, K, N, vec, bpack);
                return;
            }
        }
        hipLaunchKernelGGL((k_matvec_rows<F, NN>), dim3((unsigned)M), dim3(BLOCK), 0, st, f, A, lda, B, ldb, C, ldc, K, N, vec,
                           bvec);
    }
    template <int MM>
    static void go_vecmat(const F& f, const E* A, size_t lda, const E* B, size_t ldb, W* part, int M, int K, int N,
                          int ks, int kchunk, hipStream_t st) {
        constexpr int CW = Pack<W>::N;
        const bool vec = sizeof(E) != 12 && CW > 1 && al(B) && stride_ok(ldb) && N % CW == 0;
        if (vec) {
            dim3 grid((N / CW + BLOCK - 1) / BLOCK, ks);
            hipLaunchKernelGGL((k_vecmat_partial<F, MM, true>), grid, dim3(BLOCK), 0, st, f, A, lda, B, ldb, part, M, K, N, kchunk);
        } else {
            dim3 grid((N + BLOCK - 1) / BLOCK, ks);
            hipLaunchKernelGGL((k_vecmat_partial<F, MM, false>), grid, dim3(BLOCK), 0, st, f, A, lda, B, ldb, part, M, K, N, kchunk);
        }
    }
    static int matmul(const void* Fp, int device, const void* A, size_t lda, const void* B, size_t ldb, void* C,
                      size_t ldc, int M, int K, int N, void* workspace, size_t workspace_bytes, int mod_bits,
                      hipStream_t st) {
        const F& f = *reinterpret_cast<const F*>(Fp);
        if constexpr (F::EPW == 1) {
            // (three-limb words: the eight-column kernel would spill, N in 5..8 takes the tiled product)
            if (N <= (sizeof(W) > 16 ? 4 : SKINNY_MAX) && M >= 64 && K >= 1) {
                const E* a = (const E*)A; const E* b = (const E*)B; E* c = (E*)C;
                if (N == 1) go_matvec<1>(f, a, lda, b, ldb, c, ldc, M, K, N, st);
                else if (N == 2) go_matvec<2>(f, a, lda, b, ldb, c, ldc, M, K, N, st);
                else if (N <= 4) go_matvec<4>(f, a, lda, b, ldb, c, ldc, M, K, N, st);
                else go_matvec<8>(f, a, lda, b, ldb, c, ldc, M, K, N, st);
                FFGPU_CHECK_LAUNCH();
                return 0;
            }
            if (M <= SKINNY_MAX && N >= 64 && K >= 1 && workspace) {
                // split K so that about 2^18 threads are in flight; each chunk at least 8 rows
                const int cols_blocks = (N / (int)Pack<W>::N + BLOCK - 1) / BLOCK;
                int ks = (1024 + cols_blocks - 1) / cols_blocks;
                if (ks > (K + 7) / 8) ks = (K + 7) / 8;
                if (ks < 1) ks = 1;
                while (ks > 1 && (size_t)ks * M * N * sizeof(W) > workspace_bytes) ks /= 2;
                if ((size_t)ks * M * N * sizeof(W) <= workspace_bytes) {
                    const int kchunk = (K + ks - 1) / ks;
                    ks = (K + kchunk - 1) / kchunk;
                    W* part = (W*)workspace;
                    const E* a = (const E*)A; const E* b = (const E*)B;
                    if (M == 1) go_vecmat<1>(f, a, lda, b, ldb, part, M, K, N, ks, kchunk, st);
                    else if (M == 2) go_vecmat<2>(f, a, lda, b, ldb, part, M, K, N, ks, kchunk, st);
                    else if (M <= 4) go_vecmat<4>(f, a, lda, b, ldb, part, M, K, N, ks, kchunk, st);
                    else go_vecmat<8>(f, a, lda, b, ldb, part, M, K, N, ks, kchunk, st);
                    hipLaunchKernelGGL((k_vecmat_final<F>), dim3((unsigned)(((size_t)M * N + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0,
                                       st, f, (const W*)part, ks, M, N, (E*)C, ldc);
                    FFGPU_CHECK_LAUNCH();
                    return 0;
                }
            }
        }
        if constexpr (F::EPW == 1 && !F::BINARY && sizeof(W) <= 8) {
            // large dense products over primes of up to 64 bits: int8 matrix cores (k_limb_gemm), 8 signed base-256
            // digits per operand (4 for 32-bit storage)
            static int use_mfma = -1;
            if (use_mfma < 0) {
                const char* e = getenv("FFGPU_MM_MFMA");
                use_mfma = e ? atoi(e) : 1;
            }
            const int L = sizeof(W) == 4 ? 4 : 8;
            const uint64_t pmod = (uint64_t)f.p;
            (void)mod_bits;
            const int Mp = (M + 63) / 64 * 64, Np = (N + 63) / 64 * 64, Kp = (K + 31) / 32 * 32;
            const size_t need = (size_t)L * ((size_t)Mp + Np) * Kp;
            if (use_mfma && M >= 64 && N >= 64 && K >= 64 && (double)M * N * K >= mfma_min_macs() && workspace && need <= workspace_bytes) {
                int8_t* Ap = (int8_t*)workspace;
                int8_t* Bp = Ap + (size_t)L * Mp * Kp;
                const unsigned ga = (unsigned)(((size_t)Mp * Kp + BLOCK - 1) / BLOCK);
                dim3 gb(Np / 32, Kp / 32), gg(Np / 64, Mp / 64);
                auto go = [&](auto lc_) {
                    constexpr int LL = decltype(lc_)::value;
                    hipLaunchKernelGGL((k_limb_split_a<F, LL>), dim3(ga), dim3(BLOCK), 0, st, (const E*)A, lda, pmod, Ap, M, K, Mp, Kp);
                    // up to 128 rows: the product kernel converts B itself (BRAW), no digit planes of B
                    constexpr bool CAN_RAW = LL == 8 && sizeof(E) == 8;
                    const bool braw = CAN_RAW && gg.y <= 2 && use_mfma == 1 && limb_braw();
                    if (!braw)
                        hipLaunchKernelGGL((k_limb_split_bt<F, LL>), gb, dim3(BLOCK), 0, st, (const E*)B, ldb, pmod, Bp, K, N, Np, Kp);
                    auto product = [&](dim3 grid, E* out, size_t out_ld, int kb, int ke, int acc_, int kslice, size_t zs) {
                        if constexpr (CAN_RAW) {
                            // operand tiles straight into LDS, two k-steps ahead (k_limb_gemm_glds); raw B needs whole
                            // tiles and 16-byte aligned rows
                            const bool raw_ok = K % 32 == 0 && N % 64 == 0 && ldb % 2 == 0 && (((uintptr_t)B) & 15) == 0;
                            if (limb_glds() && (!braw || raw_ok)) {
                                if (braw) launch_glds<true>(f, grid, st, Ap, (const int8_t*)nullptr, out, out_ld, M, N, Kp, kb, ke, acc_, kslice, zs, (const E*)B, ldb, pmod);
                                else launch_glds<false>(f, grid, st, Ap, Bp, out, out_ld, M, N, Kp, kb, ke, acc_, kslice, zs, (const E*)nullptr, (size_t)0, (uint64_t)0);
                                return;
                            }
                            if (braw) {
                                hipLaunchKernelGGL((k_limb_gemm_lds<F, LL, true>), grid, dim3(BLOCK), 0, st, f, (const int8_t*)Ap,
                                                   (const int8_t*)nullptr, out, out_ld, M, N, Mp, Np, Kp, kb, ke, acc_, kslice, zs,
                                                   (const E*)B, ldb, K, pmod);
                                return;
                            }
                        }
                        hipLaunchKernelGGL((k_limb_gemm_lds<F, LL, false>), grid, dim3(BLOCK), 0, st, f, (const int8_t*)Ap,
                                           (const int8_t*)Bp, out, out_ld, M, N, Mp, Np, Kp, kb, ke, acc_, kslice, zs,
                                           (const E*)nullptr, (size_t)0, 0, (uint64_t)0);
                    };
                    // few output tiles (a batch of 64..256 rows against a big matrix): split K over blockIdx.z into
                    // slabs behind the planes, summed by k_splitk_sum
                    const size_t tiles = (size_t)gg.x * gg.y;
                    int ks = 1;
                    if (use_mfma != 2 && tiles <= 128 && Kp >= 512) {
                        // ONE round of workgroups (a workgroup holds a CU: 512 registers per lane): tiles x slabs ~ CUs.
                        // Measured (tools/mm_ks.py, 64 x 4096 x 4096): 256 workgroups 94 us, 384: 127, 512: 107, 768 (the
                        // round-2 choice): 117, 1536: 121 -- every extra slab repeats the epilogue and the pipeline fill.
                        const int ncu = launch_cfg(device).num_cu;
                        int target = ncu > 0 ? ncu : 256;
                        if (const char* e = getenv("FFGPU_MM_KS")) {          // workgroups aimed at (A/B measurements)
                            if (atoi(e) > 0) target = atoi(e);
                        }
                        ks = (int)((target + tiles - 1) / tiles);
                        if (ks > Kp / 256) ks = Kp / 256;
                        while (ks > 1 && need + 256 + (size_t)ks * M * N * sizeof(E) > workspace_bytes) --ks;
                    }
                    if (ks > 1 && Kp <= LIMB_KCHUNK) {
                        const int kslice = ((Kp + ks - 1) / ks + 31) / 32 * 32;
                        ks = (Kp + kslice - 1) / kslice;
                        E* slabs = (E*)((char*)workspace + ((need + 255) / 256) * 256);
                        dim3 g3(gg.x, gg.y, ks);
                        product(g3, slabs, (size_t)N, 0, Kp, 0, kslice, (size_t)M * N);
                        hipLaunchKernelGGL((k_splitk_sum<F>), dim3((unsigned)(((size_t)M * N + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, st, f,
                                           (const E*)slabs, ks, M, N, (E*)C, ldc);
                        return;
                    }
                    for (int kb = 0; kb < Kp; kb += LIMB_KCHUNK) {
                        const int ke = kb + LIMB_KCHUNK < Kp ? kb + LIMB_KCHUNK : Kp;
                        if (use_mfma == 2)
                            hipLaunchKernelGGL((k_limb_gemm<F, LL>), gg, dim3(BLOCK), 0, st, f, (const int8_t*)Ap, (const int8_t*)Bp,
                                               (E*)C, ldc, M, N, Mp, Np, Kp, kb, ke, kb > 0 ? 1 : 0);
                        else
                            product(gg, (E*)C, ldc, kb, ke, kb > 0 ? 1 : 0, 0, (size_t)0);
                    }
                };
                if (L == 4) go(std::integral_constant<int, 4>());
                else go(std::integral_constant<int, 8>());
                FFGPU_CHECK_LAUNCH();
                return 0;
            }
        }
        if constexpr (F::EPW == 1 && !F::BINARY && sizeof(W) == 16) {
            // primes of 65..128 bits: the matrix-core product in passes over the diagonals (k_limb_gemm_wide)
            static int use_mfma_w = -1;
            if (use_mfma_w < 0) {
                const char* e = getenv("FFGPU_MM_MFMA");
                use_mfma_w = e ? atoi(e) : 1;
            }
            constexpr int LW = sizeof(E) == 12 ? 12 : 16;
            const int Mp = (M + 63) / 64 * 64, Np = (N + 63) / 64 * 64, Kp = (K + 31) / 32 * 32;
            const size_t need = (size_t)LW * ((size_t)Mp + Np) * Kp;
            if (use_mfma_w && M >= 64 && N >= 64 && K >= 64 && (double)M * N * K >= mfma_min_macs() && workspace &&
                need <= workspace_bytes) {
                int8_t* Ap = (int8_t*)workspace;
                int8_t* Bp = Ap + (size_t)LW * Mp * Kp;
                const unsigned ga = (unsigned)(((size_t)Mp * Kp + BLOCK - 1) / BLOCK);
                dim3 gb(Np / 32, Kp / 32), gg(Np / 64, Mp / 64);
                hipLaunchKernelGGL((k_limb_split_a_wide<F, LW>), dim3(ga), dim3(BLOCK), 0, st, (const E*)A, lda, f.p_lo, f.p_hi, Ap, M,
                                   K, Mp, Kp);
                hipLaunchKernelGGL((k_limb_split_bt_wide<F, LW>), gb, dim3(BLOCK), 0, st, (const E*)B, ldb, f.p_lo, f.p_hi, Bp, K, N,
                                   Np, Kp);
                // 256^D0 mod p by repeated doubling of the canonical 1 (host, canonical arithmetic of the policy)
                auto pow256 = [&](int d0) {
                    W v;
                    v.lo = 1;
                    v.hi = 0;
                    for (int i = 0; i < 8 * d0; ++i) v = f.add(v, v);
                    return v;
                };
                bool first = true;
                for (int kb = 0; kb < Kp; kb += LIMB_KCHUNK_WIDE) {
                    const int ke = kb + LIMB_KCHUNK_WIDE < Kp ? kb + LIMB_KCHUNK_WIDE : Kp;
                    auto pass = [&](auto d0_, auto ndp_) {
                        constexpr int D0 = decltype(d0_)::value, NDP = decltype(ndp_)::value;
                        hipLaunchKernelGGL((k_limb_gemm_wide<F, LW, D0, NDP>), gg, dim3(BLOCK), 0, st, f, (const int8_t*)Ap,
                                           (const int8_t*)Bp, (E*)C, ldc, M, N, Mp, Np, Kp, kb, ke, first ? 0 : 1, pow256(D0));
                        first = false;
                    };
                    if constexpr (LW == 12) {            // 23 diagonals: 12 + 11
                        pass(std::integral_constant<int, 0>(), std::integral_constant<int, 12>());
                        pass(std::integral_constant<int, 12>(), std::integral_constant<int, 11>());
                    } else {                              // 31 diagonals: 11 + 10 + 10
                        pass(std::integral_constant<int, 0>(), std::integral_constant<int, 11>());
                        pass(std::integral_constant<int, 11>(), std::integral_constant<int, 10>());
                        pass(std::integral_constant<int, 21>(), std::integral_constant<int, 10>());
                    }
                }
                FFGPU_CHECK_LAUNCH();
                return 0;
            }
        }
        if constexpr (F::EPW > 1) {
            dim3 grid((N + 31) / 32, (M + 31) / 32);
            hipLaunchKernelGGL((k_matmul_bytes<F>), grid, dim3(BLOCK), 0, st, f, (const uint8_t*)A, lda,
                               (const uint8_t*)B, ldb, (uint8_t*)C, ldc, M, K, N);
        } else {
            static int tile = -1;
            if (tile < 0) {
                const char* e = getenv("FFGPU_MM_TILE");
                tile = e ? atoi(e) : 42;   // 4x2 per thread: measured best (1.93 T MAC/s at 4096^3 over GF(2^61-1))
            }
            // small outputs (a 64 x 64 product is two 64 x 32 tiles): 32 x 32 tiles give four times as many workgroups
            const bool small_out = ((M + 63) / 64) * ((N + 31) / 32) < 64;
            const int tcode = (sizeof(W) >= 16 || small_out) ? 22 : tile;   // two- and three-limb words: 2x2 keeps two waves per SIMD
            const int bm = tcode == 42 ? 64 : tcode == 22 ? 32 : tcode == 84 ? 128 : 64;
            const int bn = tcode == 42 ? 32 : tcode == 22 ? 32 : tcode == 84 ? 64 : 64;
            dim3 grid((N + bn - 1) / bn, (M + bm - 1) / bm);
            // too few output tiles to fill 256 CUs: split K over blockIdx.z into slabs of the workspace
            int ks = 1, kchunk = 0;
            const size_t tiles = (size_t)grid.x * grid.y;
            E* out = (E*)C;
            size_t out_ld = ldc, zstride = 0;
            if (tiles < 512 && K >= 64 && workspace) {
                ks = (int)((1024 + tiles - 1) / tiles);
                if (ks > K / 32) ks = K / 32;
                while (ks > 1 && (size_t)ks * M * N * sizeof(E) > workspace_bytes) ks /= 2;
                if (ks > 1) {
                    kchunk = ((K + ks - 1) / ks + 15) / 16 * 16;
                    ks = (K + kchunk - 1) / kchunk;
                    grid.z = ks;
                    out = (E*)workspace;
                    out_ld = N;
                    zstride = (size_t)M * N;
                }
            }
            if (ks <= 1) kchunk = 0;
            const E* a = (const E*)A; const E* b = (const E*)B;
            if (tcode == 42)
                hipLaunchKernelGGL((k_matmul<F, 4, 2>), grid, dim3(BLOCK), 0, st, f, a, lda, b, ldb, out, out_ld, M, K, N, kchunk, zstride);
            else if (tcode == 22)
                hipLaunchKernelGGL((k_matmul<F, 2, 2>), grid, dim3(BLOCK), 0, st, f, a, lda, b, ldb, out, out_ld, M, K, N, kchunk, zstride);
            else if (tcode == 84)
                hipLaunchKernelGGL((k_matmul<F, 8, 4>), grid, dim3(BLOCK), 0, st, f, a, lda, b, ldb, out, out_ld, M, K, N, kchunk, zstride);
            else
                hipLaunchKernelGGL((k_matmul<F, 4, 4>), grid, dim3(BLOCK), 0, st, f, a, lda, b, ldb, out, out_ld, M, K, N, kchunk, zstride);
            if (ks > 1)
                hipLaunchKernelGGL((k_splitk_sum<F>), dim3((unsigned)(((size_t)M * N + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, st, f,
                                   (const E*)workspace, ks, M, N, (E*)C, ldc);
        }
        FFGPU_CHECK_LAUNCH();
        return 0;
    }
    static int dot(const void* Fp, int device, const void* a, const void* b, void* out, void* workspace, size_t n,
                   hipStream_t st) {
        const F& f = *reinterpret_cast<const F*>(Fp);
        LaunchCfg lc = launch_cfg(device);
        bool vec = al(a) && (!b || al(b));
        size_t nvec = vec ? n / EPV : 0;
        size_t iters = nvec ? nvec : n;
        size_t want = (iters + (size_t)BLOCK * 8 - 1) / ((size_t)BLOCK * 8);     // >= 8 packs per thread
        unsigned grid = (unsigned)(want < 1 ? 1 : want > DOT_MAX_BLOCKS ? DOT_MAX_BLOCKS : want);
        W* part = (W*)workspace;
        if (b)
            hipLaunchKernelGGL((k_dot_partial<F, true>), dim3(grid), dim3(BLOCK), 0, st, f, (const E*)a, (const E*)b,
                               part, nvec, n);
        else
            hipLaunchKernelGGL((k_dot_partial<F, false>), dim3(grid), dim3(BLOCK), 0, st, f, (const E*)a, (const E*)a,
                               part, nvec, n);
        hipLaunchKernelGGL((k_dot_final<F>), dim3(1), dim3(BLOCK), 0, st, f, (const W*)part, (int)grid, (E*)out);
        FFGPU_CHECK_LAUNCH();
        (void)lc;
        return 0;
    }
    static int sqrt_cl(const void* Fp, int device, const void* a, const ExpArgs* eleg, const ExpArgs* elad, void* out,
                       size_t n, hipStream_t st) {
        if constexpr (F::BINARY) {
            return 2;
        } else {
            const F& f = *reinterpret_cast<const F*>(Fp);
            LaunchCfg lc = launch_cfg(device);
            unsigned grid = grid_for(n, lc);
            hipLaunchKernelGGL((k_sqrt_cl<F>), dim3(grid), dim3(BLOCK), 0, st, f, (const E*)a, *eleg, *elad, (E*)out, n);
            FFGPU_CHECK_LAUNCH();
            return 0;
        }
    }
    static int gauss(const void* Fp, int device, void* A, int n, int ncols, size_t batch, int det_mode,
                     const ExpArgs* ex, void* det, int* sing, hipStream_t st) {
        const F& f = *reinterpret_cast<const F*>(Fp);
        (void)device;
        constexpr int TI = 4;
        constexpr size_t ZMAX = 32768;                    // grid.z limit: larger batches go in chunks
        for (size_t b0 = 0; b0 < batch; b0 += ZMAX) {
            unsigned nb = (unsigned)(batch - b0 < ZMAX ? batch - b0 : ZMAX);
            E* Ab = (E*)A + b0 * (size_t)n * ncols;
            E* db = det ? (E*)det + b0 : nullptr;
            for (int k = 0; k < n; ++k) {
                hipLaunchKernelGGL((k_gauss_pivot<F>), dim3(nb), dim3(BLOCK), 0, st, f, Ab, n, ncols, k, *ex, db,
                                   sing + b0);
                int cols = ncols - k - 1;
                int rows = det_mode ? n - k - 1 : n;
                if (cols > 0 && rows > 0) {
                    dim3 grid((cols + BLOCK - 1) / BLOCK, (rows + TI - 1) / TI, nb);
                    hipLaunchKernelGGL((k_gauss_elim<F, TI>), grid, dim3(BLOCK), 0, st, f, Ab, n, ncols, k, det_mode,
                                       sing + b0);
                }
            }
        }
        FFGPU_CHECK_LAUNCH();
        return 0;
    }
    static int group_matvec(const void* Fp, int device, const uint64_t* m2, const uint64_t* bias2, int r, int g,
                            const void* in, void* out, size_t ngroups, hipStream_t st) {
        const F& f = *reinterpret_cast<const F*>(Fp);
        if (r < 1 || g < 1 || r > GM_MAX || g > GM_MAX) return 2;
        LaunchCfg lc = launch_cfg(device);
        GroupMatArgs<F> ga;
        memset(&ga, 0, sizeof(ga));
        for (int i = 0; i < r * g; ++i) ga.m[i] = f.prep(word_at<F>(f, m2, i));
        for (int a = 0; a < r; ++a) {
            W b = bias2 ? word_at<F>(f, bias2, a) : word_from_limbs<F>(f, 0, 0);
            if constexpr (F::EPW > 1) b &= 0xffu;     // one element per word on this (element-wise) path
            ga.bias[a] = b;
        }
        ga.r = r;
        ga.g = g;
        unsigned grid = grid_for(ngroups, lc);
        if constexpr (F::EPW == 4) {
            const bool al8 = (((uintptr_t)in) & 7u) == 0;
            if (g == 8 && r == 8 && al8 && (((uintptr_t)out) & 7u) == 0) {
                hipLaunchKernelGGL((k_group8_bytes<F, 8>), dim3(grid), dim3(BLOCK), 0, st, f, ga, (const uint8_t*)in,
                                   (uint8_t*)out, ngroups);
                FFGPU_CHECK_LAUNCH();
                return 0;
            }
            if (g == 8 && r == 1 && al8) {
                hipLaunchKernelGGL((k_group8_bytes<F, 1>), dim3(grid), dim3(BLOCK), 0, st, f, ga, (const uint8_t*)in,
                                   (uint8_t*)out, ngroups);
                FFGPU_CHECK_LAUNCH();
                return 0;
            }
        }
        hipLaunchKernelGGL((k_group_matvec<F>), dim3(grid), dim3(BLOCK), 0, st, f, ga, (const E*)in, (E*)out, ngroups);
        FFGPU_CHECK_LAUNCH();
        return 0;
    }
    static int beaver(const void* Fp, int device, const void* z, const void* x, const void* y, const void* d,
                      const void* e, void* out, int add_de, size_t n, hipStream_t st) {
        const F& f = *reinterpret_cast<const F*>(Fp);
        LaunchCfg lc = launch_cfg(device);
        bool vec = al(z) && al(x) && al(y) && al(d) && al(e) && al(out);
        size_t nvec = vec ? n / EPV : 0;
        unsigned grid = grid_for(nvec ? nvec : n, lc);
        hipLaunchKernelGGL((k_beaver<F, true>), dim3(grid), dim3(BLOCK), 0, st, f, (const E*)z, (const E*)x, (const E*)y,
                           (const E*)d, (const E*)e, (E*)out, add_de, nvec, n);
        FFGPU_CHECK_LAUNCH();
        return 0;
    }
    static int prss(const void* Fp, int device, const void* const* streams, int ks, int d, int l, int mask_bits,
                    const uint64_t* weights2, const uint64_t* r2, int accumulate, void* out, size_t n,
                    hipStream_t st) {
        const F& f = *reinterpret_cast<const F*>(Fp);
        LaunchCfg lc = launch_cfg(device);
        if (ks < 1 || d < 1 || l < 1 || ks > PRSS_MAXS || ks * d > PRSS_MAXW) return 2;
        PrssArgs<F> pa;
        memset(&pa, 0, sizeof(pa));
        for (int s = 0; s < ks; ++s) pa.streams[s] = (const uint8_t*)streams[s];
        for (int i = 0; i < ks * d; ++i)
            pa.w[i] = f.prep(word_at<F>(f, weights2, i));
        pa.r0 = r2[0];
        pa.r1 = r2[1];
        pa.ks = ks; pa.d = d; pa.l = l; pa.mask_bits = mask_bits; pa.accumulate = accumulate;
        unsigned grid = grid_for(n, lc);
        hipLaunchKernelGGL((k_prss<F>), dim3(grid), dim3(BLOCK), 0, st, f, pa, (E*)out, n);
        FFGPU_CHECK_LAUNCH();
        return 0;
    }

    static int prss_chacha(const void* Fp, int device, const uint8_t* keys40, int ks, int d, int l, int mask_bits, int rounds,
                           const uint64_t* weights2, const uint64_t* r2, int accumulate, void* out, size_t n, hipStream_t st) {
        const F& f = *reinterpret_cast<const F*>(Fp);
        if (ks < 1 || d < 1 || l < 1 || l > 64 || ks > PRSS_CC_MAXS || ks * d > PRSS_CC_MAXW) return 2;
        PrssCcArgs<F> pa;
        memset(&pa, 0, sizeof(pa));
        for (int s = 0; s < ks; ++s) {
            memcpy(pa.key[s], keys40 + 40 * s, 32);
            memcpy(pa.nonce[s], keys40 + 40 * s + 32, 8);
        }
        for (int i = 0; i < ks * d; ++i) pa.w[i] = f.prep(word_at<F>(f, weights2, i));
        pa.r0 = r2[0];
        pa.r1 = r2[1];
        pa.ks = ks; pa.d = d; pa.l = l; pa.mask_bits = mask_bits; pa.accumulate = accumulate; pa.rounds = rounds;
        prss_cc_layout(l, &pa.tb, &pa.dpt);
        const size_t tiles = (n + (size_t)pa.dpt - 1) / (size_t)pa.dpt;
        const unsigned grid = (unsigned)((tiles + BLOCK - 1) / BLOCK);
        hipLaunchKernelGGL((k_prss_chacha<F>), dim3(grid), dim3(BLOCK), 0, st, f, pa, (E*)out, n);
        FFGPU_CHECK_LAUNCH();
        return 0;
    }

    static const FieldOps* table() {
        static const FieldOps ops = {&ew2, &ew1, &muladd, &split, &rng_coeffs, &recombine, &pow, &inv, &matmul, &dot, &gate, &sqrt_cl, &gauss, &group_matvec, &beaver, &prss, &prss_chacha};
        return &ops;
    }
};

}  // namespace ffgpu
