// mfma_rate.hip -- issue-rate microbenchmark for the int8 matrix-core instructions in the register pattern of
// k_limb_gemm_glds / _l4: one wave per SIMD, ND resident accumulator tiles, L x L products per step, operands fixed in
// registers (no memory traffic).  Prints int8 TOP/s per variant.
//   hipcc -O3 --offload-arch=gfx950 tools/mfma_rate.hip -o build/mfma_rate
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef long v2l __attribute__((ext_vector_type(2)));

template <int L, int MODE>
__global__ __launch_bounds__(256) void k_rate(int* out, int steps, int seed) {
    constexpr int ND = 2 * L - 1;
    v4i a[L], b[L];
    // seed == 1: small constants (mostly zero bytes); otherwise pseudo-random full-range digits per lane -- the data
    // the product really sees (switching activity matters for the sustained clock)
    unsigned x = (threadIdx.x + 1u) * 2654435761u + (unsigned)seed * 40503u + blockIdx.x * 97u;
    auto rnd = [&]() { x ^= x << 13; x ^= x >> 17; x ^= x << 5; return (int)x; };
    for (int l = 0; l < L; ++l) {
        if (seed == 1) {
            a[l] = (v4i){seed + l, seed * 3 + l, seed ^ l, seed + 7 * l};
            b[l] = (v4i){seed - l, seed * 5 + l, seed ^ (l << 3), seed + 11 * l};
        } else {
            a[l] = (v4i){rnd(), rnd(), rnd(), rnd()};
            b[l] = (v4i){rnd(), rnd(), rnd(), rnd()};
        }
    }
    if (MODE == 0) {                       // 32x32x32, all 2L-1 diagonals resident (the kernel's pattern)
        v16i acc[ND];
        for (int d = 0; d < ND; ++d) acc[d] = (v16i){0};
        for (int s = 0; s < steps; ++s) {
#pragma unroll
            for (int la = 0; la < L; ++la)
#pragma unroll
                for (int lb = 0; lb < L; ++lb)
                    acc[la + lb] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[la], b[lb], acc[la + lb], 0, 0, 0);
        }
        int r = 0;
        for (int d = 0; d < ND; ++d) for (int q = 0; q < 16; ++q) r ^= acc[d][q];
        out[blockIdx.x * 256 + threadIdx.x] = r;
    } else if (MODE == 1) {                // 32x32x32, L*L independent-ish: only 4 accumulators round robin
        v16i acc[4];
        for (int d = 0; d < 4; ++d) acc[d] = (v16i){0};
        for (int s = 0; s < steps; ++s) {
#pragma unroll
            for (int i = 0; i < L * L; ++i)
                acc[i & 3] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[i % L], b[(i / L) % L], acc[i & 3], 0, 0, 0);
        }
        int r = 0;
        for (int d = 0; d < 4; ++d) for (int q = 0; q < 16; ++q) r ^= acc[d][q];
        out[blockIdx.x * 256 + threadIdx.x] = r;
    } else {                               // 16x16x64: same MAC count per step = 2 * L * L instructions, 8 accumulators
        typedef int v4acc __attribute__((ext_vector_type(4)));
        v4acc acc[8];
        for (int d = 0; d < 8; ++d) acc[d] = (v4acc){0};
        for (int s = 0; s < steps; ++s) {
#pragma unroll
            for (int i = 0; i < 2 * L * L; ++i)
                acc[i & 7] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[i % L], b[(i / L) % L], acc[i & 7], 0, 0, 0);
        }
        int r = 0;
        for (int d = 0; d < 8; ++d) for (int q = 0; q < 4; ++q) r ^= acc[d][q];
        out[blockIdx.x * 256 + threadIdx.x] = r;
    }
}

static int g_seed = 1;
template <int L, int MODE>
static void run(const char* name, int* out, int wgs) {
    const int steps = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k_rate<L, MODE>), dim3(wgs), dim3(256), 0, 0, out, 10, 1);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k_rate<L, MODE>), dim3(wgs), dim3(256), 0, 0, out, steps, g_seed);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double macs = (double)wgs * 4 * steps * L * L * 32768.0;
    printf("%-44s wgs %4d  %8.3f ms  %8.1f int8 TOP/s\n", name, wgs, ms, 2 * macs / (ms * 1e-3) / 1e12);
}
int main() {
    int* out;
    hipMalloc(&out, 4096 * 256 * 4);
    for (int pass = 0; pass < 2; ++pass) {
      g_seed = pass == 0 ? 1 : 12345;
      printf("operands: %s\n", pass == 0 ? "small constants" : "random full-range int8 digits");
      for (int wgs : {256, 512}) {
        run<8, 0>("32x32x32 i8, 15 resident diagonals (L=8)", out, wgs);
        run<8, 1>("32x32x32 i8, 4 accumulators round robin", out, wgs);
        run<8, 2>("16x16x64 i8, 8 accumulators round robin", out, wgs);
        run<4, 0>("32x32x32 i8, 7 resident diagonals (L=4)", out, wgs);
      }
    }
    return 0;
}
