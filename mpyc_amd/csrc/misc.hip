// misc.hip -- kernels that are not templated on a field policy:
//   * the GF(2^8) S-box layer on local/public bytes (demos/np_aes.py:37-43):
//     a byte->byte map, so the 256-entry table is built once per call on the
//     host (t^254 by the reference's addition chain, runtime.py:1356-1367,
//     then the GF(2) affine map), handed over as a kernel argument, copied to
//     LDS by every workgroup, and 16 bytes per lane stream through 16 look-ups;
//   * the streaming copy used as the achievable-HBM-bandwidth yardstick.
#include <stdlib.h>
#include "kernels.hpp"

using namespace ffgpu;

struct SboxLut {
    uint8_t v[256];
};

FF_HD uint32_t gf_pow254(const GF2P8& f, uint32_t a) {
    uint32_t d = a;
    uint32_t c = f.mul(d, d);  // a^2
    c = f.mul(c, c);           // a^4
    c = f.mul(c, c);           // a^8
    c = f.mul(c, d);           // a^9
    c = f.mul(c, c);           // a^18
    uint32_t c2 = f.mul(c, c); // a^36
    d = f.mul(c, d);           // a^19
    c = c2;
    c2 = f.mul(c, c);          // a^72
    d = f.mul(c, d);           // a^55
    c = f.mul(c2, d);          // a^127
    return f.mul(c, c);        // a^254
}

static int table_blocks_per_cu() {
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("FFGPU_TABLE_BPC");
        v = e ? atoi(e) : 0;  // 0 = uncapped: the per-workgroup table copy is cheap (measured best)
    }
    return v;
}

__global__ __launch_bounds__(BLOCK) void k_sbox(SboxLut tab, const uint8_t* __restrict__ in,
                                                 uint8_t* __restrict__ out, size_t nvec, size_t n) {
    __shared__ uint8_t lut[256];
    lut[threadIdx.x] = tab.v[threadIdx.x];  // BLOCK == 256: one table entry per thread
    __syncthreads();
    const size_t gid = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    const size_t gsz = (size_t)gridDim.x * BLOCK;
    const uint4* __restrict__ iv = reinterpret_cast<const uint4*>(in);
    uint4* __restrict__ ov = reinterpret_cast<uint4*>(out);
    for (size_t i = gid; i < nvec; i += gsz) {
        uint4 v = ldg<true>(iv + i);
        uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            uint32_t x = w[q];
            w[q] = (uint32_t)lut[x & 0xff] | ((uint32_t)lut[(x >> 8) & 0xff] << 8) |
                   ((uint32_t)lut[(x >> 16) & 0xff] << 16) | ((uint32_t)lut[x >> 24] << 24);
        }
        stg<true>(ov + i, make_uint4(w[0], w[1], w[2], w[3]));
    }
    for (size_t e = nvec * 16 + gid; e < n; e += gsz) out[e] = lut[in[e]];
}

// host: the byte -> byte map, built with the same packed GF(2^8) arithmetic the element-wise kernels use:
// t^254 by the reference's addition chain (runtime.py:1356-1367), then the GF(2) affine map.
// (api.hip caches it per context: it only depends on (rows8, b).)
int ffgpu_sbox_build_lut(const void* policy, const uint8_t* rows8, uint8_t b, uint8_t* lut256) {
    const GF2P8& f = *reinterpret_cast<const GF2P8*>(policy);
    for (uint32_t t = 0; t < 256; ++t) {
        uint32_t inv = gf_pow254(f, t) & 0xffu;
        uint32_t y = 0;
        for (int r = 0; r < 8; ++r) y |= (uint32_t)(__builtin_popcount(inv & rows8[r]) & 1) << r;
        lut256[t] = (uint8_t)(y ^ b);
    }
    return 0;
}

int ffgpu_launch_sbox(const uint8_t* lut256, int device, const void* in, void* out, size_t n, hipStream_t st) {
    SboxLut tab;
    memcpy(tab.v, lut256, 256);
    LaunchCfg lc = launch_cfg(device);
    bool vec = aligned16(in) && aligned16(out);
    size_t nvec = vec ? n / 16 : 0;
    LaunchCfg capped = lc;
    if (capped.blocks_per_cu <= 0) capped.blocks_per_cu = table_blocks_per_cu();
    unsigned grid = grid_for(nvec ? nvec : n, capped);
    hipLaunchKernelGGL(k_sbox, dim3(grid), dim3(BLOCK), 0, st, tab, (const uint8_t*)in, (uint8_t*)out, nvec, n);
    FFGPU_CHECK_LAUNCH();
    return 0;
}

// ---- GF(2^n), n <= 8: multiplication through log / antilog tables in LDS ---------------------
// c = exp[log a + log b].  log[0] = 2*(q-1) so that any sum involving a zero operand lands in the
// zero-padded tail of exp[] (no zero test, no select).  Both tables come from the host (built at
// context creation with the same packed shift-xor arithmetic) as a kernel argument; each workgroup
// copies them to LDS once (1.5 KiB) and then streams 16 bytes per lane.
// log: 256 x u16 (512 B), exp: 4*(q-1)+1 <= 1021 x u8 -> 1.5 KiB of LDS, (nearly) conflict-free.
struct Gf8Tables {
    uint16_t lg[256];
    uint8_t ex[1024];
};

__global__ __launch_bounds__(BLOCK) void k_gf8_mul_tab(Gf8Tables tb, const uint8_t* __restrict__ a,
                                                        const uint8_t* __restrict__ b, uint8_t* __restrict__ out,
                                                        size_t nvec, size_t n) {
    __shared__ uint16_t lg[256];
    __shared__ uint8_t ex[1024];
    {
        uint32_t t = threadIdx.x;
        lg[t] = tb.lg[t];
        reinterpret_cast<uint32_t*>(ex)[t] = reinterpret_cast<const uint32_t*>(tb.ex)[t];
    }
    __syncthreads();
    const size_t gid = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    const size_t gsz = (size_t)gridDim.x * BLOCK;
    const uint4* __restrict__ av = reinterpret_cast<const uint4*>(a);
    const uint4* __restrict__ bv = reinterpret_cast<const uint4*>(b);
    uint4* __restrict__ ov = reinterpret_cast<uint4*>(out);
    for (size_t i = gid; i < nvec; i += gsz) {
        uint4 x = ldg<true>(av + i);
        uint4 y = ldg<true>(bv + i);
        uint32_t xw[4] = {x.x, x.y, x.z, x.w}, yw[4] = {y.x, y.y, y.z, y.w}, r[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            uint32_t acc = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                uint32_t s = (uint32_t)lg[(xw[q] >> (8 * k)) & 0xff] + (uint32_t)lg[(yw[q] >> (8 * k)) & 0xff];
                acc |= (uint32_t)ex[s] << (8 * k);
            }
            r[q] = acc;
        }
        stg<true>(ov + i, make_uint4(r[0], r[1], r[2], r[3]));
    }
    for (size_t e = nvec * 16 + gid; e < n; e += gsz) out[e] = ex[(uint32_t)lg[a[e]] + (uint32_t)lg[b[e]]];
}

// host: build the tables for GF(2^n) (any irreducible modulus of degree n <= 8)
int ffgpu_gf8_build_tables(const void* policy, void* tables_out) {
    const GF2P8& f = *reinterpret_cast<const GF2P8*>(policy);
    Gf8Tables* tb = reinterpret_cast<Gf8Tables*>(tables_out);
    const uint32_t q1 = (1u << f.n) - 1;  // multiplicative order
    memset(tb, 0, sizeof(*tb));
    if (q1 == 1) {  // GF(2): 1*1 = 1
        tb->lg[0] = 2;
        tb->lg[1] = 0;
        tb->ex[0] = 1;
        return 0;
    }
    for (uint32_t g = 2; g <= q1; ++g) {
        // order of g
        uint32_t x = 1, ord = 0;
        do {
            x = f.mul(x, g) & 0xffu;
            ++ord;
        } while (x != 1 && ord <= q1);
        if (ord != q1) continue;
        x = 1;
        for (uint32_t i = 0; i < q1; ++i) {
            tb->ex[i] = (uint8_t)x;
            tb->ex[i + q1] = (uint8_t)x;  // sums of two logs reach 2*(q1-1)
            tb->lg[x] = (uint16_t)i;
            x = f.mul(x, g) & 0xffu;
        }
        tb->lg[0] = (uint16_t)(2 * q1);   // 2*q1 .. 4*q1 stay zero in ex[]
        for (uint32_t v = q1 + 1; v < 256; ++v) tb->lg[v] = (uint16_t)(2 * q1);  // non-canonical bytes -> 0
        return 0;
    }
    return 1;
}

int ffgpu_launch_gf8_mul_tab(const void* tables, int device, const void* a, const void* b, void* out, size_t n,
                             hipStream_t st) {
    const Gf8Tables& tb = *reinterpret_cast<const Gf8Tables*>(tables);
    LaunchCfg lc = launch_cfg(device);
    bool vec = aligned16(a) && aligned16(b) && aligned16(out);
    size_t nvec = vec ? n / 16 : 0;
    LaunchCfg capped = lc;
    if (capped.blocks_per_cu <= 0) capped.blocks_per_cu = table_blocks_per_cu();
    unsigned grid = grid_for(nvec ? nvec : n, capped);
    hipLaunchKernelGGL(k_gf8_mul_tab, dim3(grid), dim3(BLOCK), 0, st, tb, (const uint8_t*)a, (const uint8_t*)b,
                       (uint8_t*)out, nvec, n);
    FFGPU_CHECK_LAUNCH();
    return 0;
}

__global__ __launch_bounds__(BLOCK) void k_copy16(const uint4* __restrict__ src, uint4* __restrict__ dst,
                                                   size_t nvec) {
    const size_t gid = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    const size_t gsz = (size_t)gridDim.x * BLOCK;
    for (size_t i = gid; i < nvec; i += gsz) {
        uint4 x = ldg<true>(src + i);
        stg<true>(dst + i, x);
    }
}

int ffgpu_launch_copy(int device, const void* src, void* dst, size_t bytes, hipStream_t st) {
    if (!aligned16(src) || !aligned16(dst) || (bytes & 15)) return 1;
    LaunchCfg lc = launch_cfg(device);
    size_t nvec = bytes / 16;
    unsigned grid = grid_for(nvec, lc);
    hipLaunchKernelGGL(k_copy16, dim3(grid), dim3(BLOCK), 0, st, (const uint4*)src, (uint4*)dst, nvec);
    FFGPU_CHECK_LAUNCH();
    return 0;
}
