// matmul.hpp -- products with a matrix operand: the LDS-tiled dense product on the VALU (k_matmul), the skinny shapes of
// the Vandermonde / activation-row products, and the exact modular GEMM on the int8 matrix cores (signed base-256 digit
// planes; the one GEMM-shaped piece of the path).  Included by kernels.hpp.
#pragma once

namespace ffgpu {

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
template <class F, bool LZ>
struct MatmulDigits {                      // digits per staged element when the policy accumulates in 28-bit digits (round 6)
    enum { NL = 1 };
};
template <class F>
struct MatmulDigits<F, true> {
    enum { NL = F::LAZY_NL };
};
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
    // multi-limb 2^k - c primes (round 6): the tiles are staged as 28-bit DIGITS and every term is NL^2 multiply-adds into
    // column sums (fields.hpp LazyDot), reduced every 32 terms -- ~100 instructions per term with the 128-bit limb arithmetic
    constexpr bool LZ = HasLazyAcc<F>::value;
    constexpr int NL = MatmulDigits<F, LZ>::NL;
    constexpr int BK = 16, BM = 16 * TM, BN = 16 * TN, FLUSH = LZ ? (int)FF_D28_MAX_TERMS : 192;
    static_assert(FLUSH % BK == 0, "the flush test follows whole k-steps");
    using Acc = typename std::conditional<LZ, typename LazyAccOf<F>::type, typename F::acc>::type;
    __shared__ W As[LZ ? 1 : BK][LZ ? 1 : BM + 1];
    __shared__ W Bs[LZ ? 1 : BK][LZ ? 1 : BN + 1];
    __shared__ uint32_t Ad[LZ ? BK : 1][LZ ? BM + 1 : 1][NL];
    __shared__ uint32_t Bd[LZ ? BK : 1][LZ ? BN + 1 : 1][NL];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    Acc acc[TM][TN];
    W tot[TM][TN];
    bool have = false;
    auto zero = [&](Acc& a_) {
        if constexpr (LZ) f.lacc_zero(a_); else f.acc_zero(a_);
    };
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) zero(acc[i][j]);
    int since = 0;
    for (int k0 = 0; k0 < K; k0 += BK) {
        // stage A (BM x BK) transposed and B (BK x BN)
        for (int idx = threadIdx.x; idx < BM * BK; idx += BLOCK) {
            int mm = idx / BK, kk = idx % BK;
            int gm = m0 + mm, gk = k0 + kk;
            const bool ok = gm < M && gk < K;      // out-of-range: read element 0, then zero it
            const W v = ff_keep_if<W>(f.prep(ld_elem<F>(A, ok ? (size_t)gm * lda + gk : 0)), ok);
            if constexpr (LZ) {
                uint32_t d[NL];
                f.lacc_digits(v, d);
#pragma unroll
                for (int t_ = 0; t_ < NL; ++t_) Ad[kk][mm][t_] = d[t_];
            } else {
                As[kk][mm] = v;
            }
        }
        for (int idx = threadIdx.x; idx < BK * BN; idx += BLOCK) {
            int kk = idx / BN, nn = idx % BN;
            int gk = k0 + kk, gn = n0 + nn;
            const bool ok = gk < K && gn < N;
            const W v = ff_keep_if<W>(ld_elem<F>(B, ok ? (size_t)gk * ldb + gn : 0), ok);
            if constexpr (LZ) {
                uint32_t d[NL];
                f.lacc_digits(v, d);
#pragma unroll
                for (int t_ = 0; t_ < NL; ++t_) Bd[kk][nn][t_] = d[t_];
            } else {
                Bs[kk][nn] = v;
            }
        }
        __syncthreads();
#pragma unroll 4
        for (int kk = 0; kk < BK; ++kk) {
            if constexpr (LZ) {
                uint32_t a[TM][NL], b[TN][NL];
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int t_ = 0; t_ < NL; ++t_) a[i][t_] = Ad[kk][ty + 16 * i][t_];
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int t_ = 0; t_ < NL; ++t_) b[j][t_] = Bd[kk][tx + 16 * j][t_];
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) f.lacc_mac_digits(acc[i][j], a[i], b[j]);
            } else {
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
        }
        __syncthreads();
        since += BK;
        if (since >= FLUSH) {   // keep the unreduced accumulators inside their headroom (2^8 products; digit columns: 32)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    W part;
                    if constexpr (LZ) part = f.lacc_reduce(acc[i][j]); else part = f.acc_reduce(acc[i][j]);
                    tot[i][j] = have ? f.add(tot[i][j], part) : part;
                    zero(acc[i][j]);
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
                W r;
                if constexpr (LZ) r = f.lacc_reduce(acc[i][j]); else r = f.acc_reduce(acc[i][j]);
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

// 8 x 8 byte transposes for the in-kernel digit conversion of a raw right operand (k_limb_gemm_glds<BRAW>)
__device__ __forceinline__ void transpose4x4_bytes(uint32_t r0, uint32_t r1, uint32_t r2, uint32_t r3, uint32_t (&o)[4]) {
    const uint32_t t0 = __builtin_amdgcn_perm(r1, r0, 0x05010400u), t1 = __builtin_amdgcn_perm(r1, r0, 0x07030602u);
    const uint32_t t2 = __builtin_amdgcn_perm(r3, r2, 0x05010400u), t3 = __builtin_amdgcn_perm(r3, r2, 0x07030602u);
    o[0] = __builtin_amdgcn_perm(t2, t0, 0x05040100u);
    o[1] = __builtin_amdgcn_perm(t2, t0, 0x07060302u);
    o[2] = __builtin_amdgcn_perm(t3, t1, 0x05040100u);
    o[3] = __builtin_amdgcn_perm(t3, t1, 0x07060302u);
}

// Primes below 2^32 (32-bit storage, L = 4 digits, 7 diagonals): the operand tiles of a workgroup (64 x 64 outputs, four
// waves) are staged through LDS, double buffered, one tile ahead in registers -- 2 x 4 x 2 KiB per k-step.  LDS layout
// [plane][k-half][row][16 bytes]: the 16 lanes a ds_read_b128 phase serves read 256 contiguous bytes (conflict-free).
// (64-bit storage goes through k_limb_gemm_glds below, which streams the tiles into LDS without registers.)
template <class F>
__global__ __launch_bounds__(BLOCK) void k_limb_gemm_l4(F f, const int8_t* __restrict__ Ap, const int8_t* __restrict__ Bp,
                                                         typename F::elem* __restrict__ C, size_t ldc, int M, int N, int Kp,
                                                         int kb, int ke, int accumulate, int kslice, size_t zstride) {
    typedef typename F::word W;
    constexpr int L = 4, ND = 2 * L - 1;
    constexpr int CHUNKS = L * 2 * 64;                 // 16-byte chunks of one operand tile per k-step
    constexpr int PER_THREAD = CHUNKS / BLOCK;
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
    // epilogue: sum_d 256^d D_d mod p on signed diagonal sums, Horner from the top diagonal (the only place the modulus enters)
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
        const int col = bn0 + wn + (lane & 31), row = bm0 + wm + (q & 3) + 8 * (q >> 2) + 4 * (lane >> 5);
        if (row < M && col < N) {
            W v = res[q];
            if (accumulate) v = f.add(v, ld_elem<F>(C, (size_t)row * ldc + col));
            st_elem<F>(C, (size_t)row * ldc + col, v);
        }
    }
}

// ---- the same product with the operand tiles streamed STRAIGHT into LDS, three stages deep (round 4) ------------
// A register-staged tile fetch (one k-step = 64 MFMAs = ~1 us ahead) does not cover the HBM / L2 latency at one wave per
// SIMD (the 15 accumulator tiles take the register file: 512 of 512) -- measured in round 3: the operand fetch cost 37 % of
// the 4096^3 product, and the 64-row shape ran at 110 us against 31 us of MFMA work.  Here the digit-plane tiles (byte for byte their LDS image: limb_off) and, for BRAW, the
// raw rows of B go global -> LDS without passing through registers (global_load_lds_dwordx4: LDS address = wave-uniform
// base + lane * 16), TWO tiles ahead in a ring of three stages; the waves wait with a COUNTED vmcnt (the tile issued
// last stays in flight across the barrier) and synchronise with raw s_barrier (a __syncthreads would drain vmcnt to
// 0).  BRAW: the raw 64-bit elements of tile s + 1 are converted to digit bytes LDS -> LDS (same arithmetic as the
// register variant) while the MFMAs of tile s run.  No staging registers (32 fewer).  Requires K padded to 32 (planes:
// always) and, for BRAW, K % 32 == 0, N % 64 == 0 and 16-byte aligned rows of B; the launcher goes through digit
// planes of B otherwise.
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
// (the column-sum kernels below test the count AFTER adding a whole group of terms: a flush happens at no more than
// SKINNY_FLUSH - 1 + group terms, which must stay within ColAcc::MAX_TERMS = 256 -- asserted where each group size is known)

// The accumulator of the matrix x few-columns kernels: the policy's dot product in 28-bit digits where it has one (the
// multi-limb 2^k - c primes, round 6: ~45 instead of ~100 instructions per term, flushed every 32 terms), else F::acc.
template <class F>
struct SkinnyAcc {
    static constexpr bool LZ = HasLazyAcc<F>::value;
    typedef typename std::conditional<LZ, typename LazyAccOf<F>::type, typename F::acc>::type T;
    enum { FLUSH = LZ ? (int)FF_D28_MAX_TERMS : (int)SKINNY_FLUSH };
    static __device__ __forceinline__ void zero(const F& f, T& a) {
        if constexpr (LZ) f.lacc_zero(a); else f.acc_zero(a);
    }
    static __device__ __forceinline__ void mac(const F& f, T& a, const typename F::word& l, const typename F::word& x) {
        if constexpr (LZ) f.lacc_mac(a, l, x); else f.acc_mac(a, l, x);
    }
    static __device__ __forceinline__ typename F::word reduce(const F& f, const T& a) {
        if constexpr (LZ) return f.lacc_reduce(a); else return f.acc_reduce(a);
    }
};

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
    typedef SkinnyAcc<F> SA;
    typename SA::T acc[NN];
    W total[NN];
    bool have = false;
    int cnt = 0;
#pragma unroll
    for (int j = 0; j < NN; ++j) SA::zero(f, acc[j]);
    auto flush = [&]() {
#pragma unroll
        for (int j = 0; j < NN; ++j) {
            W part = SA::reduce(f, acc[j]);
            total[j] = have ? f.add(total[j], part) : part;
            SA::zero(f, acc[j]);
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
                        if (jp * P::N + q < NN) SA::mac(f, acc[jp * P::N + q], xp, bp.w[q]);
                }
            }
        } else {
#pragma unroll
            for (int j = 0; j < NN; ++j)
                if (j < N) SA::mac(f, acc[j], xp, ld_elem<F>(B, kk * ldb + j));
        }
        if (++cnt >= SA::FLUSH) flush();
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
    typedef SkinnyAcc<F> SA;
    typename SA::T acc[R][NN];
    W total[R][NN];
    bool have = false;
    int cnt = 0;
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int j = 0; j < NN; ++j) SA::zero(f, acc[r][j]);
    auto flush = [&]() {
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int j = 0; j < NN; ++j) {
                W part = SA::reduce(f, acc[r][j]);
                total[r][j] = have ? f.add(total[r][j], part) : part;
                SA::zero(f, acc[r][j]);
            }
        have = true;
        cnt = 0;
    };
    constexpr int EPV = P::N;
    const int nvec = vec ? K / EPV : 0;
    // (UN packs of every row and of B in flight before the first multiply-add: one pack at a time ran at the memory latency --
    // 16 dependent round trips for a 4096-element row of 12-byte elements: 53 us = 0.47 of HBM for 4096^2)
    constexpr int UN = (SA::LZ && EPV == 1) ? 4 : 1;
    auto packs = [&](int i0, auto un) {
        constexpr int U = decltype(un)::value;
        P x[U][R];
        W b[U][EPV][NN];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = i0 + u * BLOCK;
#pragma unroll
            for (int r = 0; r < R; ++r)                               // rows past M re-read the last row (result discarded)
                x[u][r] = ldg<true>(reinterpret_cast<const MP*>(A + (row0 + r < (size_t)M ? row0 + r : (size_t)M - 1) * lda) + i);
            if (bpack) {
                const P bp = ldg<false>(reinterpret_cast<const MP*>(B) + i);
#pragma unroll
                for (int q = 0; q < EPV; ++q) b[u][q][0] = f.prep(bp.w[q]);
            } else {
#pragma unroll
                for (int q = 0; q < EPV; ++q)
#pragma unroll
                    for (int j = 0; j < NN; ++j)
                        if (j < N) b[u][q][j] = f.prep(ld_elem<F>(B, ((size_t)i * EPV + q) * ldb + j));
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
#pragma unroll
            for (int r = 0; r < R; ++r)
#pragma unroll
                for (int q = 0; q < EPV; ++q)
#pragma unroll
                    for (int j = 0; j < NN; ++j)
                        if (j < N) SA::mac(f, acc[r][j], b[u][q][j], x[u][r].w[q]);
            cnt += EPV;
            if (cnt >= SA::FLUSH) flush();
        }
    };
    int i = threadIdx.x;
    if constexpr (UN > 1)
        for (; i + (UN - 1) * BLOCK < nvec; i += UN * BLOCK) packs(i, std::integral_constant<int, UN>());
    for (; i < nvec; i += BLOCK) packs(i, std::integral_constant<int, 1>());
    for (int kk = nvec * EPV + threadIdx.x; kk < K; kk += BLOCK) {
#pragma unroll
        for (int j = 0; j < NN; ++j)
            if (j < N) {
                const W bp = f.prep(ld_elem<F>(B, (size_t)kk * ldb + j));
#pragma unroll
                for (int r = 0; r < R; ++r)
                    SA::mac(f, acc[r][j], bp, ld_elem<F>(A, (row0 + r < (size_t)M ? row0 + r : (size_t)M - 1) * lda + kk));
            }
        if (++cnt >= SA::FLUSH) flush();
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

// Matrix x few columns (2 <= N <= 8) over one-word primes (col_mac_ok): column sums (fields.hpp ColAcc).  Here no operand is
// shared between lanes (every lane walks its own k), so the entry of A is split into its three limbs in the lane -- four
// instructions, used for all N columns -- and each term is six multiply-adds instead of the 32 instructions of acc_mac
// (measured over 2^61 - 1, 16384 x 4096 @ 4096 x N, round-4 kernel: N = 2 132 us, N = 3 383, N = 4 473, N = 8 1415 us against
// 85 us for reading A once).  One instantiation per N.
//
// k_matvec_sub_col (B contiguous, the usual case): SIXTEEN lanes per row of A, 16 rows per workgroup.  With a whole
// workgroup per row (below) a thread sees K / 256 terms and then pays ~200 instructions per column for the reduction and the
// butterfly over the workgroup -- as much as its share of the product at K = 4096 -- and every 16-byte load of B by a wave
// touches 64 different cache lines.  Here a lane sees K / 16 terms, the butterfly has four steps inside the wave, and the 512
// rows of B of one tile are copied to LDS with whole 16-byte chunks; the four row groups of a wave read the same addresses
// (broadcast), and the two rows of a lane start an ODD number of chunks after its neighbour's (N odd: N, N even: N + 1)
// so that 16 neighbouring lanes fall on 16 different bank groups.
// LW lanes per row: 16, or 64 (four rows per workgroup) when there are too few rows to fill the chip with 16 per workgroup.
constexpr int MATVEC_SUB_KT = 512, MATVEC_SUB_U = 4;
template <class F, int NN, int LW>
__global__ __launch_bounds__(BLOCK) void k_matvec_sub_col(F f, const typename F::elem* __restrict__ A, size_t lda,
                                                           const typename F::elem* __restrict__ B,
                                                           typename F::elem* __restrict__ C, size_t ldc, int M, int K) {
    typedef Pack<typename F::word> P;
    typedef typename MemPack<F>::type MP;
    typedef typename F::word W;
    static_assert(P::N == 2, "two rows of B per lane and step");
    constexpr int ROWS = BLOCK / LW, KT = MATVEC_SUB_KT, U = MATVEC_SUB_U;
    constexpr int CH = (NN % 2) ? NN : NN + 1;                     // 16-byte chunks per pair of rows of B in the tile
    __shared__ __attribute__((aligned(16))) W sb[(KT / 2) * CH * 2];
    const int g = threadIdx.x / LW, l = threadIdx.x % LW;
    const size_t row = (size_t)blockIdx.x * ROWS + g;
    const typename F::elem* __restrict__ a = A + (row < (size_t)M ? row : (size_t)M - 1) * lda;   // rows past M re-read the last one
    ColAcc<typename F::acc> acc[NN];
    // two terms per step, tested after the step, + the last column of an odd K
    static_assert(SKINNY_FLUSH - 1 + 2 + 1 <= ColAcc<typename F::acc>::MAX_TERMS, "column sums overflow before the flush");
    int cnt = 0;
#pragma unroll
    for (int j = 0; j < NN; ++j) acc[j].zero();
    auto flush = [&]() {                                           // reduce, the residue re-enters as one term
#pragma unroll
        for (int j = 0; j < NN; ++j) {
            const W part = f.acc_reduce(acc[j].gather());
            acc[j].zero();
            acc[j].c00 = (uint32_t)part;
            acc[j].c01 = (uint32_t)(part >> 32);
        }
        cnt = 1;
    };
    auto term = [&](W x, const W* b) {
        const ColLimbs xl = col_limbs(f.prep(x));
#pragma unroll
        for (int j = 0; j < NN; ++j) acc[j].mac(xl, b[j]);
    };
    const int keven = K & ~1;
    for (int kt0 = 0; kt0 < keven; kt0 += KT) {
        const int pairs = (keven - kt0 < KT ? keven - kt0 : KT) / 2;
        __syncthreads();                                            // the previous tile has been read
        const MP* __restrict__ src = reinterpret_cast<const MP*>(B + (size_t)kt0 * NN);
        for (int c = threadIdx.x; c < pairs * NN; c += BLOCK)
            *reinterpret_cast<P*>(&sb[((c / NN) * CH + c % NN) * 2]) = ldg<false>(src + c);
        __syncthreads();
        const MP* __restrict__ av = reinterpret_cast<const MP*>(a + kt0);
        for (int p0 = l; p0 < pairs; p0 += LW * U) {                // U packs of A in flight per lane
            P x[U];
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (p0 + u * LW < pairs) x[u] = ldg<true>(av + p0 + u * LW);
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int p = p0 + u * LW;
                if (p < pairs) {
                    W w[2 * NN];
#pragma unroll
                    for (int c = 0; c < NN; ++c) {
                        const P v = *reinterpret_cast<const P*>(&sb[(p * CH + c) * 2]);
                        w[2 * c] = v.w[0];
                        w[2 * c + 1] = v.w[1];
                    }
                    term(x[u].w[0], w);
                    term(x[u].w[1], w + NN);
                    cnt += 2;
                    if (cnt >= SKINNY_FLUSH) flush();
                }
            }
        }
    }
    if ((K & 1) && l == 0) {                                        // the last column of an odd K
        W b[NN];
#pragma unroll
        for (int j = 0; j < NN; ++j) b[j] = ld_elem<F>(B, (size_t)(K - 1) * NN + j);
        term(ld_elem<F>(a, K - 1), b);                             // (at most SKINNY_FLUSH + 2 terms now: asserted above)
    }
#pragma unroll
    for (int j = 0; j < NN; ++j) {
        W v = f.acc_reduce(acc[j].gather());
#pragma unroll
        for (int off = LW / 2; off > 0; off >>= 1) v = f.add(v, wave_shfl_xor(v, off));
        if (l == 0 && row < (size_t)M) st_elem<F>(C, row * ldc + j, v);
    }
}

// The same for any layout of B (row stride, alignment): one workgroup per R rows of A, the rows of B read from global memory
template <class F, int NN, int R>
__global__ __launch_bounds__(BLOCK) void k_matvec_rows_col(F f, const typename F::elem* __restrict__ A, size_t lda,
                                                            const typename F::elem* __restrict__ B, size_t ldb,
                                                            typename F::elem* __restrict__ C, size_t ldc, int M, int K, int vec,
                                                            int bvec) {
    typedef Pack<typename F::word> P;
    typedef typename MemPack<F>::type MP;
    typedef typename F::word W;
    constexpr int EPV = P::N;
    __shared__ W sm[R * NN * (BLOCK / 64)];
    const size_t row0 = (size_t)blockIdx.x * R;
    ColAcc<typename F::acc> acc[R][NN];
    static_assert(SKINNY_FLUSH - 1 + EPV <= ColAcc<typename F::acc>::MAX_TERMS, "column sums overflow before the flush");
    int cnt = 0;
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int j = 0; j < NN; ++j) acc[r][j].zero();
    auto flush = [&]() {                                           // reduce, the residue re-enters as one term
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int j = 0; j < NN; ++j) {
                const W part = f.acc_reduce(acc[r][j].gather());
                acc[r][j].zero();
                acc[r][j].c00 = (uint32_t)part;
                acc[r][j].c01 = (uint32_t)(part >> 32);
            }
        cnt = 1;
    };
    auto term = [&](const W (&x)[R], const W (&b)[NN]) {           // one k: R entries of A against a row of B
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const ColLimbs xl = col_limbs(f.prep(x[r]));
#pragma unroll
            for (int j = 0; j < NN; ++j) acc[r][j].mac(xl, b[j]);
        }
    };
    auto load_b = [&](size_t kk, W (&b)[NN]) {                     // row kk of B
        bool packed = false;
        if constexpr (NN % P::N == 0) {
            if (bvec) {
                const MP* __restrict__ br = reinterpret_cast<const MP*>(B + kk * ldb);
#pragma unroll
                for (int jp = 0; jp < NN / P::N; ++jp) {
                    const P bp = ldg<false>(br + jp);
#pragma unroll
                    for (int q = 0; q < P::N; ++q) b[jp * P::N + q] = bp.w[q];
                }
                packed = true;
            }
        }
        if (!packed) {
#pragma unroll
            for (int j = 0; j < NN; ++j) b[j] = ld_elem<F>(B, kk * ldb + j);
        }
    };
    auto row_of = [&](int r) { return row0 + r < (size_t)M ? row0 + r : (size_t)M - 1; };   // rows past M re-read the last one
    const int nvec = vec ? K / EPV : 0;
    for (int i = threadIdx.x; i < nvec; i += BLOCK) {
        P x[R];
#pragma unroll
        for (int r = 0; r < R; ++r) x[r] = ldg<true>(reinterpret_cast<const MP*>(A + row_of(r) * lda) + i);
#pragma unroll
        for (int q = 0; q < EPV; ++q) {
            W xr[R], b[NN];
#pragma unroll
            for (int r = 0; r < R; ++r) xr[r] = x[r].w[q];
            load_b((size_t)i * EPV + q, b);
            term(xr, b);
        }
        cnt += EPV;
        if (cnt >= SKINNY_FLUSH) flush();
    }
    for (int kk = nvec * EPV + threadIdx.x; kk < K; kk += BLOCK) {
        W xr[R], b[NN];
#pragma unroll
        for (int r = 0; r < R; ++r) xr[r] = ld_elem<F>(A, row_of(r) * lda + kk);
        load_b((size_t)kk, b);
        term(xr, b);
        if (++cnt >= SKINNY_FLUSH) flush();
    }
    // R*NN sums over the workgroup: butterfly inside each wave, then one exchange of the per-wave sums through LDS
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int j = 0; j < NN; ++j) {
            W v = f.acc_reduce(acc[r][j].gather());
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
        if (row0 + r < (size_t)M) st_elem<F>(C, (row0 + r) * ldc + j, v);
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
// partial[(ks * M + m) * N + j], summed by k_vecmat_final.  The entries of A are the same for every thread: with STAGE the
// workgroup converts (prep) a tile of VECMAT_KT rows of A once into LDS and every lane reads it back as a broadcast -- without it
// they are scalar loads in the loop, one dependent wait per row of A (measured over 2^61 - 1 at 4096 x 4096: M = 8 126 us,
// M = 2 37.6 us against 30.5 us for M = 1), and Montgomery fields repeat the conversion in every thread.
constexpr int VECMAT_KT = 32;
// Policies with a dot product in 28-bit digits (the multi-limb 2^k - c primes, fields.hpp LazyDot) take it here too (round 6):
// the staged entries of A are stored as digits, a row of B is cut once per thread and used for all M rows, every term is NL^2
// multiply-adds into column sums, reduced every FF_D28_MAX_TERMS terms -- 8 x 4096 @ 4096^2 over the 80-bit prime: 384 us with
// the 128-bit limb arithmetic of acc_mac (~100 instructions per term).
template <class F, bool LZ>
struct VecmatDigits {                      // (only named when LZ)
    enum { NL = 1 };
};
template <class F>
struct VecmatDigits<F, true> {
    enum { NL = F::LAZY_NL };
};
template <class F, int MM, bool VEC, bool STAGE, int UNR = 4>
__global__ __launch_bounds__(BLOCK) void k_vecmat_partial(F f, const typename F::elem* __restrict__ A, size_t lda,
                                                           const typename F::elem* __restrict__ B, size_t ldb,
                                                           typename F::word* __restrict__ partial, int M, int K, int N,
                                                           int kchunk) {
    typedef Pack<typename F::word> P;
    typedef typename MemPack<F>::type MP;
    typedef typename F::word W;
    constexpr int CW = VEC ? P::N : 1;                              // columns per thread
    constexpr int KT = VECMAT_KT;
    constexpr bool LZ = HasLazyAcc<F>::value;                      // digit accumulators (staged A: its digits live in LDS)
    constexpr int NL = VecmatDigits<F, LZ>::NL;
    constexpr int FLUSH = LZ ? (int)FF_D28_MAX_TERMS : (int)SKINNY_FLUSH;
    static_assert(!LZ || FF_D28_MAX_TERMS % UNR == 0, "the flush test follows whole groups");
    using Acc = typename std::conditional<LZ, typename LazyAccOf<F>::type, typename F::acc>::type;
    __shared__ W sa[(STAGE && !LZ) ? MM : 1][(STAGE && !LZ) ? KT : 1];
    __shared__ uint32_t sad[(LZ && STAGE) ? MM : 1][(LZ && STAGE) ? KT : 1][NL];
    const int j = (blockIdx.x * BLOCK + threadIdx.x) * CW;
    const bool live = j < N;
    if (!STAGE && !live) return;
    const int k0 = blockIdx.y * kchunk;
    const int k1 = k0 + kchunk < K ? k0 + kchunk : K;
    Acc acc[MM][CW];
    W total[MM][CW];
    bool have = false;
    int cnt = 0;
    auto zero = [&](Acc& a_) {
        if constexpr (LZ) f.lacc_zero(a_); else f.acc_zero(a_);
    };
#pragma unroll
    for (int mi = 0; mi < MM; ++mi)
#pragma unroll
        for (int q = 0; q < CW; ++q) zero(acc[mi][q]);
    auto flush = [&]() {
#pragma unroll
        for (int mi = 0; mi < MM; ++mi)
#pragma unroll
            for (int q = 0; q < CW; ++q) {
                W part;
                if constexpr (LZ) part = f.lacc_reduce(acc[mi][q]); else part = f.acc_reduce(acc[mi][q]);
                total[mi][q] = have ? f.add(total[mi][q], part) : part;
                zero(acc[mi][q]);
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
    for (int kt0 = k0; kt0 < k1; kt0 += KT) {
        const int kt1 = !STAGE ? k1 : (kt0 + KT < k1 ? kt0 + KT : k1);
        if constexpr (STAGE) {
            __syncthreads();                                        // the previous tile has been read
            for (int e = threadIdx.x; e < MM * KT; e += BLOCK) {
                const int mi = e / KT, kk = kt0 + e % KT;
                W v = W();
                if (mi < M && kk < kt1) v = f.prep(ld_elem<F>(A, (size_t)mi * lda + kk));
                if constexpr (LZ) {
                    uint32_t d[NL];
                    f.lacc_digits(v, d);
#pragma unroll
                    for (int t_ = 0; t_ < NL; ++t_) sad[mi][e % KT][t_] = d[t_];
                } else {
                    sa[mi][e % KT] = v;
                }
            }
            __syncthreads();
        }
        auto macs = [&](int kk, const W (&b)[CW]) {
            if constexpr (LZ) {
                uint32_t db[CW][NL];
#pragma unroll
                for (int q = 0; q < CW; ++q) f.lacc_digits(b[q], db[q]);
#pragma unroll
                for (int mi = 0; mi < MM; ++mi) {
                    if (MM == 1 || mi < M) {
                        uint32_t da[NL];
                        if constexpr (STAGE) {
#pragma unroll
                            for (int t_ = 0; t_ < NL; ++t_) da[t_] = sad[mi][kk - kt0][t_];     // broadcast reads
                        } else {
                            f.lacc_digits(f.prep(ld_elem<F>(A, (size_t)mi * lda + kk)), da);    // wave-uniform operand: scalar unit
                        }
#pragma unroll
                        for (int q = 0; q < CW; ++q) f.lacc_mac_digits(acc[mi][q], da, db[q]);
                    }
                }
            } else {
#pragma unroll
                for (int mi = 0; mi < MM; ++mi) {
                    if (MM == 1 || mi < M) {
                        W ap;
                        if constexpr (STAGE) ap = sa[mi][kk - kt0];                                 // broadcast read
                        else ap = f.prep(ld_elem<F>(A, (size_t)mi * lda + kk));                     // wave-uniform operand
#pragma unroll
                        for (int q = 0; q < CW; ++q) f.acc_mac(acc[mi][q], ap, b[q]);
                    }
                }
            }
        };
        if (live) {
            int kk = kt0;
            for (; kk + UNR <= kt1; kk += UNR) {              // UNR rows of B in flight per thread
                W b[UNR][CW];
#pragma unroll
                for (int u = 0; u < UNR; ++u) load_b(kk + u, b[u]);
#pragma unroll
                for (int u = 0; u < UNR; ++u) macs(kk + u, b[u]);
                cnt += UNR;
                if (cnt >= FLUSH) flush();
            }
            for (; kk < kt1; ++kk) {
                W b0[CW];
                load_b(kk, b0);
                macs(kk, b0);
                if (++cnt >= FLUSH) flush();
            }
        }
        if constexpr (!STAGE) break;
    }
    if (!live) return;
    flush();
#pragma unroll
    for (int mi = 0; mi < MM; ++mi)
        if (mi < M)
#pragma unroll
            for (int q = 0; q < CW; ++q)
                if (j + q < N) partial[((size_t)blockIdx.y * M + mi) * N + j + q] = total[mi][q];
}

// The same for prime fields on one 64-bit word (col_mac_ok: PM64<*>, RC64): the staged entries of A are split into three limbs
// and every term costs six multiply-adds into column sums (fields.hpp ColAcc) instead of a 128-bit product plus a 192-bit
// carry chain.  One instantiation per M (no test inside the row loop), one column per thread (a wave reads 512 contiguous
// bytes of a row of B; two columns per thread double the registers of the column sums and were never faster).  The rows of B
// are read in groups of UNR, the NEXT group in flight while the current one is multiplied: with the loads issued and awaited
// inside one iteration the kernel ran at the memory latency (M = 8: 55-66 us, the multiply-adds at half their issue rate).
constexpr int VECMAT_COL_KT = 128;                                  // rows of A staged at once (16 B each per row of A)
template <class F, int MM, int UNR, int MINB>
__global__ __launch_bounds__(BLOCK, MINB) void k_vecmat_partial_col(F f, const typename F::elem* __restrict__ A, size_t lda,
                                                               const typename F::elem* __restrict__ B, size_t ldb,
                                                               typename F::word* __restrict__ partial, int K, int N, int kchunk) {
    typedef typename F::word W;
    constexpr int KT = VECMAT_COL_KT;
    __shared__ ColLimbs sa[MM][KT];
    const int j = blockIdx.x * BLOCK + threadIdx.x;
    const bool live = j < N;
    const int k0 = blockIdx.y * kchunk;
    const int k1 = k0 + kchunk < K ? k0 + kchunk : K;
    ColAcc<typename F::acc> acc[MM];
    static_assert(SKINNY_FLUSH - 1 + UNR <= ColAcc<typename F::acc>::MAX_TERMS, "column sums overflow before the flush");
    int cnt = 0;
#pragma unroll
    for (int mi = 0; mi < MM; ++mi) acc[mi].zero();
    // every SKINNY_FLUSH terms: reduce, and put the residue back as one term (1 x residue: its two words are column sums)
    auto flush = [&]() {
#pragma unroll
        for (int mi = 0; mi < MM; ++mi) {
            const W part = f.acc_reduce(acc[mi].gather());
            acc[mi].zero();
            acc[mi].c00 = (uint32_t)part;
            acc[mi].c01 = (uint32_t)(part >> 32);
        }
        cnt = 1;
    };
    const typename F::elem* Bj = B + (live ? j : 0);
    auto load_group = [&](int kk, W (&b)[UNR]) {
#pragma unroll
        for (int u = 0; u < UNR; ++u) b[u] = ld_elem<F>(Bj, (size_t)(kk + u) * ldb);
    };
    for (int kt0 = k0; kt0 < k1; kt0 += KT) {
        const int kt1 = kt0 + KT < k1 ? kt0 + KT : k1;
        __syncthreads();                                            // the previous tile has been read
        for (int e = threadIdx.x; e < MM * KT; e += BLOCK) {
            const int mi = e / KT, kk = kt0 + e % KT;
            if (kk < kt1) sa[mi][e % KT] = col_limbs(f.prep(ld_elem<F>(A, (size_t)mi * lda + kk)));
        }
        __syncthreads();
        if (!live) continue;
        auto mac_group = [&](int kk, const W (&b)[UNR]) {
#pragma unroll
            for (int u = 0; u < UNR; ++u)
#pragma unroll
                for (int mi = 0; mi < MM; ++mi) acc[mi].mac(sa[mi][kk + u - kt0], b[u]);        // broadcast reads
            cnt += UNR;
            if (cnt >= SKINNY_FLUSH) flush();
        };
        int kk = kt0;
        if (kk + UNR <= kt1) {
            W ba[UNR], bb[UNR];
            load_group(kk, ba);
            for (;;) {                                              // ping-pong: the group after the current one is in flight
                const bool more = kk + 2 * UNR <= kt1;
                if (more) load_group(kk + UNR, bb);
                mac_group(kk, ba);
                kk += UNR;
                if (!more) break;
                const bool more2 = kk + 2 * UNR <= kt1;
                if (more2) load_group(kk + UNR, ba);
                mac_group(kk, bb);
                kk += UNR;
                if (!more2) break;
            }
        }
        for (; kk < kt1; ++kk) {
            const W b0 = ld_elem<F>(Bj, (size_t)kk * ldb);
#pragma unroll
            for (int mi = 0; mi < MM; ++mi) acc[mi].mac(sa[mi][kk - kt0], b0);
            if (++cnt >= SKINNY_FLUSH) flush();
        }
    }
    if (!live) return;
#pragma unroll
    for (int mi = 0; mi < MM; ++mi) partial[((size_t)blockIdx.y * MM + mi) * N + j] = f.acc_reduce(acc[mi].gather());
}

// Sum of the KS partial rows.  M x N is small (4096 outputs for an activation row against 4096^2 weights) and KS ~ 128: one thread
// per output is sixteen blocks walking a chain of loads.  Eight threads share an output instead (thread group g takes the
// partials s = g mod 8, four loads in flight), then one pass through LDS.
constexpr int VECMAT_FINAL_G = 8;
template <class F>
__global__ __launch_bounds__(BLOCK) void k_vecmat_final(F f, const typename F::word* __restrict__ partial, int KS, int M,
                                                         int N, typename F::elem* __restrict__ C, size_t ldc) {
    typedef typename F::word W;
    constexpr int G = VECMAT_FINAL_G, CO = BLOCK / G;               // outputs per block
    __shared__ W sm[G][CO];
    const int c = threadIdx.x % CO, g = threadIdx.x / CO;
    const size_t idx = (size_t)blockIdx.x * CO + c;
    const size_t mn = (size_t)M * N;
    const bool live = idx < mn;
    if (live && g < KS) {
        int s = g;
        W r = partial[(size_t)s * mn + idx];
        s += G;
        for (; s + 3 * G < KS; s += 4 * G) {
            const W t0 = partial[(size_t)s * mn + idx], t1 = partial[(size_t)(s + G) * mn + idx];
            const W t2 = partial[(size_t)(s + 2 * G) * mn + idx], t3 = partial[(size_t)(s + 3 * G) * mn + idx];
            r = f.add(r, f.add(f.add(t0, t1), f.add(t2, t3)));
        }
        for (; s < KS; s += G) r = f.add(r, partial[(size_t)s * mn + idx]);
        sm[g][c] = r;
    }
    __syncthreads();
    if (g == 0 && live) {
        W r = sm[0][c];
        for (int q = 1; q < G && q < KS; ++q) r = f.add(r, sm[q][c]);
        const int mi = (int)(idx / N), j = (int)(idx % N);
        st_elem<F>(C, (size_t)mi * ldc + j, r);
    }
}

}  // namespace ffgpu
