// misc.hip -- kernels that are not templated on a field policy:
//   * the GF(2^8) S-box layer on local/public bytes (demos/np_aes.py:37-43):
//     a byte->byte map, so each workgroup builds the 256-entry table once in
//     LDS (thread t computes entry t with the same packed GF(2^8) arithmetic
//     as the element-wise kernels: t^254 by the reference's addition chain,
//     runtime.py:1356-1367, then the GF(2) affine map) and then streams
//     16 bytes per lane through 16 LDS look-ups;
//   * the streaming copy used as the achievable-HBM-bandwidth yardstick.
#include "kernels.hpp"

using namespace ffgpu;

struct SboxArgs {
    uint8_t rows[8];
    uint8_t b;
};

__device__ __forceinline__ uint32_t gf_pow254(const GF2P8& f, uint32_t a) {
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

__global__ __launch_bounds__(BLOCK) void k_sbox(GF2P8 f, SboxArgs sa, const uint8_t* __restrict__ in,
                                                 uint8_t* __restrict__ out, size_t nvec, size_t n) {
    __shared__ uint8_t lut[256];
    {
        uint32_t t = threadIdx.x;  // BLOCK == 256: one table entry per thread
        uint32_t inv = gf_pow254(f, t) & 0xffu;
        uint32_t y = 0;
#pragma unroll
        for (int r = 0; r < 8; ++r) y |= (uint32_t)(__popc(inv & sa.rows[r]) & 1) << r;
        lut[t] = (uint8_t)(y ^ sa.b);
    }
    __syncthreads();
    const size_t gid = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    const size_t gsz = (size_t)gridDim.x * BLOCK;
    const uint4* __restrict__ iv = reinterpret_cast<const uint4*>(in);
    uint4* __restrict__ ov = reinterpret_cast<uint4*>(out);
    for (size_t i = gid; i < nvec; i += gsz) {
        uint4 v = iv[i];
        uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            uint32_t x = w[q];
            w[q] = (uint32_t)lut[x & 0xff] | ((uint32_t)lut[(x >> 8) & 0xff] << 8) |
                   ((uint32_t)lut[(x >> 16) & 0xff] << 16) | ((uint32_t)lut[x >> 24] << 24);
        }
        ov[i] = make_uint4(w[0], w[1], w[2], w[3]);
    }
    for (size_t e = nvec * 16 + gid; e < n; e += gsz) out[e] = lut[in[e]];
}

int ffgpu_launch_sbox(const void* policy, int device, const void* in, const uint8_t* rows8, uint8_t b,
                      void* out, size_t n, hipStream_t st) {
    const GF2P8& f = *reinterpret_cast<const GF2P8*>(policy);
    SboxArgs sa;
    for (int r = 0; r < 8; ++r) sa.rows[r] = rows8[r];
    sa.b = b;
    LaunchCfg lc = launch_cfg(device);
    bool vec = aligned16(in) && aligned16(out);
    size_t nvec = vec ? n / 16 : 0;
    // every workgroup builds the 256-entry table first (~1.4k VALU ops per thread), so unlike the
    // pure streaming kernels this one runs as a persistent grid: 8 workgroups per CU, grid-stride.
    LaunchCfg capped = lc;
    if (capped.blocks_per_cu <= 0) capped.blocks_per_cu = 8;
    unsigned grid = grid_for(nvec ? nvec : n, capped);
    hipLaunchKernelGGL(k_sbox, dim3(grid), dim3(BLOCK), 0, st, f, sa, (const uint8_t*)in, (uint8_t*)out, nvec,
                       n);
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
