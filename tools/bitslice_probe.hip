// tools/bitslice_probe.hip -- GF(2^64) element-wise product with BIT-SLICED arithmetic (measurement probe, not in the
// product).  VERDICT r3 "next #4": hold 32 elements per lane as 64 bit-planes of uint32, one partial-product MAC =
// one v_bitop3_b32 (acc ^ (a & b)), Karatsuba down to 8 x 8 leaves, sparse-modulus fold, 32 x 32 bit transposes in and out.
//   hipcc -O3 --offload-arch=gfx950 -save-temps -c tools/bitslice_probe.hip   -> static VALU count / registers (tools/isa_hist.py,
//                                                                                tools/kernel_regs.py)
//   hipcc -O3 --offload-arch=gfx950 tools/bitslice_probe.hip -o bsp && ./bsp   -> parity against a bit-serial product + timing
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define DEV __device__ __forceinline__
DEV uint32_t mac(uint32_t acc, uint32_t a, uint32_t b) { return __builtin_amdgcn_bitop3_b32(acc, a, b, 0x78); }   // acc ^ (a & b)
DEV uint32_t xor3(uint32_t a, uint32_t b, uint32_t c) { return __builtin_amdgcn_bitop3_b32(a, b, c, 0x96); }
DEV uint32_t bsel(uint32_t m, uint32_t a, uint32_t b) { return __builtin_amdgcn_bitop3_b32(m, a, b, 0xCA); }      // m ? a : b

// 32 x 32 bit-matrix transpose in registers: out word i, bit e = in word e, bit i
template <int S>
DEV void tr_stage(uint32_t (&A)[32]) {
    constexpr uint32_t M = S == 16 ? 0x0000ffffu : S == 8 ? 0x00ff00ffu : S == 4 ? 0x0f0f0f0fu : S == 2 ? 0x33333333u : 0x55555555u;
#pragma unroll
    for (int k = 0; k < 32; ++k) {
        if (k & S) continue;
        const uint32_t x = A[k], y = A[k + S];
        if constexpr (S == 16) {
            A[k] = __builtin_amdgcn_perm(y, x, 0x05040100u);          // lo16(x) | lo16(y) << 16
            A[k + S] = __builtin_amdgcn_perm(y, x, 0x07060302u);      // hi16(x) | hi16(y) << 16
        } else if constexpr (S == 8) {
            A[k] = __builtin_amdgcn_perm(y, x, 0x06020400u);          // bytes x0, y0, x2, y2
            A[k + S] = __builtin_amdgcn_perm(y, x, 0x07030501u);      // bytes x1, y1, x3, y3
        } else {
            A[k] = bsel(M, x, y << S);
            A[k + S] = bsel(M, x >> S, y);
        }
    }
}
DEV void transpose32(uint32_t (&A)[32]) {
    tr_stage<16>(A); tr_stage<8>(A); tr_stage<4>(A); tr_stage<2>(A); tr_stage<1>(A);
}

// c (2N - 1 planes) = a (N planes) x b (N planes) over GF(2)[x], bit-sliced
template <int N>
struct BsMul {
    static DEV void run(const uint32_t* a, const uint32_t* b, uint32_t* c) {
        constexpr int H = N / 2;
        uint32_t z0[N - 1], z2[N - 1], zm[N - 1], am[H], bm[H];
        BsMul<H>::run(a, b, z0);
        BsMul<H>::run(a + H, b + H, z2);
#pragma unroll
        for (int i = 0; i < H; ++i) { am[i] = a[i] ^ a[i + H]; bm[i] = b[i] ^ b[i + H]; }
        BsMul<H>::run(am, bm, zm);
#pragma unroll
        for (int k = 0; k < 2 * N - 1; ++k) {
            uint32_t v = k < N - 1 ? z0[k] : (k >= N ? z2[k - N] : 0u);
            const int q = k - H;
            if (q >= 0 && q < N - 1) {
                const uint32_t mid = xor3(zm[q], z0[q], z2[q]);
                v = (k == N - 1) ? mid : (v ^ mid);
            }
            c[k] = v;
        }
    }
};
template <>
struct BsMul<8> {
    static DEV void run(const uint32_t* a, const uint32_t* b, uint32_t* c) {
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

// x^64 + x^4 + x^3 + x + 1
DEV void fold64(uint32_t (&c)[127]) {
#pragma unroll
    for (int k = 126; k >= 64; --k) {
        const uint32_t h = c[k];
        c[k - 64] ^= h; c[k - 63] ^= h; c[k - 61] ^= h; c[k - 60] ^= h;
    }
}

template <int WAVES>      // 1: 256 VGPRs + 64 AGPRs, one wave per SIMD;  2: held to 256 registers (64 words spill to scratch)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WAVES, WAVES))) void k_bs_mul64(const uint4* __restrict__ a, const uint4* __restrict__ b, uint4* __restrict__ o,
                                                  size_t nslab) {
    // a wave owns 2048 consecutive elements = 1024 uint4; lane l reads uint4 number r * 64 + l, r = 0..15 (coalesced)
    const size_t wave = ((size_t)blockIdx.x * 256 + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    if (wave >= nslab) return;
    const size_t base = wave * 1024 + lane;
    uint32_t pa[64], pb[64];
    {
        uint32_t lo[32], hi[32];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const uint4 v = a[base + (size_t)r * 64];
            lo[2 * r] = v.x; hi[2 * r] = v.y; lo[2 * r + 1] = v.z; hi[2 * r + 1] = v.w;
        }
        transpose32(lo); transpose32(hi);
#pragma unroll
        for (int i = 0; i < 32; ++i) { pa[i] = lo[i]; pa[32 + i] = hi[i]; }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const uint4 v = b[base + (size_t)r * 64];
            lo[2 * r] = v.x; hi[2 * r] = v.y; lo[2 * r + 1] = v.z; hi[2 * r + 1] = v.w;
        }
        transpose32(lo); transpose32(hi);
#pragma unroll
        for (int i = 0; i < 32; ++i) { pb[i] = lo[i]; pb[32 + i] = hi[i]; }
    }
    uint32_t c[127];
    BsMul<64>::run(pa, pb, c);
    fold64(c);
    uint32_t lo[32], hi[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) { lo[i] = c[i]; hi[i] = c[32 + i]; }
    transpose32(lo); transpose32(hi);
#pragma unroll
    for (int r = 0; r < 16; ++r) o[base + (size_t)r * 64] = make_uint4(lo[2 * r], hi[2 * r], lo[2 * r + 1], hi[2 * r + 1]);
}

// variant 3: ONE wave per SIMD, persistent over the slabs, the NEXT slab's 32 loads in flight while the current one is
// multiplied (128 more registers -- there are 512 at this occupancy)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void k_bs_mul64_pipe(const uint4* __restrict__ a,
                                                                                                const uint4* __restrict__ b,
                                                                                                uint4* __restrict__ o, size_t nslab) {
    const size_t wave0 = ((size_t)blockIdx.x * 256 + threadIdx.x) >> 6;
    const size_t nwaves = ((size_t)gridDim.x * 256) >> 6;
    const int lane = threadIdx.x & 63;
    if (wave0 >= nslab) return;
    uint4 na[16], nb[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) { na[r] = a[wave0 * 1024 + lane + (size_t)r * 64]; nb[r] = b[wave0 * 1024 + lane + (size_t)r * 64]; }
    for (size_t wave = wave0; wave < nslab; wave += nwaves) {
        const size_t base = wave * 1024 + lane;
        uint32_t pa[64], pb[64];
        {
            uint32_t lo[32], hi[32];
#pragma unroll
            for (int r = 0; r < 16; ++r) { lo[2 * r] = na[r].x; hi[2 * r] = na[r].y; lo[2 * r + 1] = na[r].z; hi[2 * r + 1] = na[r].w; }
            transpose32(lo); transpose32(hi);
#pragma unroll
            for (int i = 0; i < 32; ++i) { pa[i] = lo[i]; pa[32 + i] = hi[i]; }
#pragma unroll
            for (int r = 0; r < 16; ++r) { lo[2 * r] = nb[r].x; hi[2 * r] = nb[r].y; lo[2 * r + 1] = nb[r].z; hi[2 * r + 1] = nb[r].w; }
            transpose32(lo); transpose32(hi);
#pragma unroll
            for (int i = 0; i < 32; ++i) { pb[i] = lo[i]; pb[32 + i] = hi[i]; }
        }
        const size_t nxt = wave + nwaves;
        if (nxt < nslab) {
#pragma unroll
            for (int r = 0; r < 16; ++r) { na[r] = a[nxt * 1024 + lane + (size_t)r * 64]; nb[r] = b[nxt * 1024 + lane + (size_t)r * 64]; }
        }
        uint32_t c[127];
        BsMul<64>::run(pa, pb, c);
        fold64(c);
        uint32_t lo[32], hi[32];
#pragma unroll
        for (int i = 0; i < 32; ++i) { lo[i] = c[i]; hi[i] = c[32 + i]; }
        transpose32(lo); transpose32(hi);
#pragma unroll
        for (int r = 0; r < 16; ++r) o[base + (size_t)r * 64] = make_uint4(lo[2 * r], hi[2 * r], lo[2 * r + 1], hi[2 * r + 1]);
    }
}

static uint64_t ref_mul64(uint64_t a, uint64_t b) {
    uint64_t lo = 0, hi = 0;
    for (int i = 0; i < 64; ++i)
        if ((b >> i) & 1) { lo ^= a << i; if (i) hi ^= a >> (64 - i); }
    for (int k = 63; k >= 0; --k)
        if ((hi >> k) & 1) {
            const int taps[4] = {0, 1, 3, 4};
            for (int t : taps) {
                const int pos = k + t;
                if (pos >= 64) hi ^= 1ull << (pos - 64); else lo ^= 1ull << pos;
            }
        }
    return lo;
}

int main() {
    const size_t n = 10'000'000 / 2048 * 2048 + 2048;
    std::vector<uint64_t> ha(n), hb(n), ho(n);
    uint64_t s = 88172645463325252ull;
    auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; };
    for (size_t i = 0; i < n; ++i) { ha[i] = rnd(); hb[i] = rnd(); }
    ha[0] = 0; ha[1] = 1; hb[1] = ~0ull; ha[2] = ~0ull; hb[2] = ~0ull;
    void *da, *db, *dout;
    hipMalloc(&da, n * 8); hipMalloc(&db, n * 8); hipMalloc(&dout, n * 8);
    hipMemcpy(da, ha.data(), n * 8, hipMemcpyHostToDevice);
    hipMemcpy(db, hb.data(), n * 8, hipMemcpyHostToDevice);
    const size_t nslab = n / 2048;
    const unsigned grid = (unsigned)((nslab * 64 + 255) / 256);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    size_t bad_total = 0;
    for (int variant = 1; variant <= 4; ++variant) {
        auto launch = [&]() {
            if (variant == 1) k_bs_mul64<1><<<grid, 256>>>((const uint4*)da, (const uint4*)db, (uint4*)dout, nslab);
            else if (variant == 2) k_bs_mul64<2><<<grid, 256>>>((const uint4*)da, (const uint4*)db, (uint4*)dout, nslab);
            else k_bs_mul64_pipe<<<variant == 3 ? 256 : 512, 256>>>((const uint4*)da, (const uint4*)db, (uint4*)dout, nslab);   // 1 (2) blocks per CU
        };
        hipMemset(dout, 0, n * 8);
        for (int rep = 0; rep < 3; ++rep) launch();
        hipEventRecord(e0);
        const int reps = 20;
        for (int rep = 0; rep < reps; ++rep) launch();
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(ho.data(), dout, n * 8, hipMemcpyDeviceToHost);
        size_t bad = 0;
        for (size_t i = 0; i < n; i += (i < 4096 ? 1 : 997)) bad += ho[i] != ref_mul64(ha[i], hb[i]);
        printf("bit-sliced GF(2^64) product, variant %d (1, 2: waves per SIMD; 3, 4: persistent + prefetch, 256 / 512 blocks): n=%zu  %.1f us per launch  %.1f GB/s of 24 B/element (%.3f of 8 TB/s)  "
               "mismatches=%zu\n", variant, n, ms / reps * 1e3, 24.0 * n / (ms / reps * 1e-3) / 1e9, 24.0 * n / (ms / reps * 1e-3) / 8e12, bad);
        bad_total += bad;
    }
    const size_t bad = bad_total;
    return bad != 0;
}
