// mfma_limb_gemm.hip -- PROTOTYPE: exact modular GEMM over GF(2^61-1) on the int8 matrix cores.
// Operands are split into nine 7-bit limbs (int8 planes, K-contiguous: A planes [l][M][K], B planes
// TRANSPOSED [l][N][K]); for every pair of limbs v_mfma_i32_32x32x32_i8 accumulates into the i32 accumulator of
// the diagonal d = la + lb (17 diagonals; 9 * 127^2 * K < 2^31 for K <= 14795); the epilogue evaluates
// sum_d D_d 2^(7d) mod p.  One wave = one 32x32 output tile, 4 waves per workgroup (64x64).
// hipcc --offload-arch=gfx950 -O3 tools/mfma_limb_gemm.hip -o build/mfma_limb_gemm
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef unsigned __int128 u128;
static const uint64_t P = (1ull << 61) - 1;
enum { L = 9, ND = 2 * L - 1 };

__global__ void k_split_a(const uint64_t* __restrict__ A, int8_t* __restrict__ Ap, int M, int K) {
    size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (size_t)M * K) return;
    uint64_t v = A[idx];
    for (int l = 0; l < L; ++l) Ap[(size_t)l * M * K + idx] = (int8_t)((v >> (7 * l)) & 127);
}
// B (K x N) -> planes [l][N][K]
__global__ void k_split_bt(const uint64_t* __restrict__ B, int8_t* __restrict__ Bp, int K, int N) {
    __shared__ uint64_t tile[32][33];
    int n0 = blockIdx.x * 32, k0 = blockIdx.y * 32;
    int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;      // 256 threads: 8 rows per pass
    for (int r = ty; r < 32; r += 8) tile[r][tx] = B[(size_t)(k0 + r) * N + n0 + tx];   // tile[k][n]
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {                     // r = n, tx = k
        uint64_t v = tile[tx][r];
        for (int l = 0; l < L; ++l) Bp[(size_t)l * N * K + (size_t)(n0 + r) * K + k0 + tx] = (int8_t)((v >> (7 * l)) & 127);
    }
}

__device__ __forceinline__ uint64_t mod61(u128 x) {
    uint64_t lo = (uint64_t)x & P, hi = (uint64_t)(x >> 61);
    uint64_t s = lo + hi;                 // hi < 2^67: fold twice
    s = (s & P) + (s >> 61);
    return s >= P ? s - P : s;
}

__global__ __launch_bounds__(256) void k_gemm(const int8_t* __restrict__ Ap, const int8_t* __restrict__ Bp,
                                              uint64_t* __restrict__ C, int M, int K, int N) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int m0 = blockIdx.y * 64 + (wave >> 1) * 32, n0 = blockIdx.x * 64 + (wave & 1) * 32;
    const int r = lane & 31, h = lane >> 5;
    v16i acc[ND];
#pragma unroll
    for (int d = 0; d < ND; ++d) acc[d] = (v16i){0};
    const size_t planeA = (size_t)M * K, planeB = (size_t)N * K;
    const int8_t* pa = Ap + (size_t)(m0 + r) * K + 16 * h;
    const int8_t* pb = Bp + (size_t)(n0 + r) * K + 16 * h;
    for (int k0 = 0; k0 < K; k0 += 32) {
        v4i a[L], b[L];
#pragma unroll
        for (int l = 0; l < L; ++l) {
            a[l] = *reinterpret_cast<const v4i*>(pa + l * planeA + k0);
            b[l] = *reinterpret_cast<const v4i*>(pb + l * planeB + k0);
        }
#pragma unroll
        for (int la = 0; la < L; ++la)
#pragma unroll
            for (int lb = 0; lb < L; ++lb)
                acc[la + lb] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[la], b[lb], acc[la + lb], 0, 0, 0);
    }
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const int col = n0 + (lane & 31), row = m0 + (q & 3) + 8 * (q >> 2) + 4 * (lane >> 5);
        uint64_t res = 0;
#pragma unroll
        for (int d = ND - 1; d >= 0; --d) res = mod61(((u128)res << 7) + (uint32_t)acc[d][q]);
        C[(size_t)row * N + col] = res;
    }
}

int main(int argc, char** argv) {
    int M = argc > 1 ? atoi(argv[1]) : 1024, K = argc > 2 ? atoi(argv[2]) : 1024, N = argc > 3 ? atoi(argv[3]) : 1024;
    std::vector<uint64_t> hA((size_t)M * K), hB((size_t)K * N);
    srand(7);
    auto rnd = []() { uint64_t v = ((uint64_t)rand() << 42) ^ ((uint64_t)rand() << 21) ^ rand(); return v % P; };
    for (auto& v : hA) v = rnd();
    for (auto& v : hB) v = rnd();
    for (int i = 0; i < K; ++i) { hA[i] = P - 1; hB[(size_t)i * N] = P - 1; }     // worst case in C[0][0]
    uint64_t *dA, *dB, *dC; int8_t *dAp, *dBp;
    hipMalloc(&dA, hA.size() * 8); hipMalloc(&dB, hB.size() * 8); hipMalloc(&dC, (size_t)M * N * 8);
    hipMalloc(&dAp, (size_t)L * M * K); hipMalloc(&dBp, (size_t)L * N * K);
    hipMemcpy(dA, hA.data(), hA.size() * 8, hipMemcpyHostToDevice);
    hipMemcpy(dB, hB.data(), hB.size() * 8, hipMemcpyHostToDevice);
    hipEvent_t e0, e1, e2; hipEventCreate(&e0); hipEventCreate(&e1); hipEventCreate(&e2);
    auto run = [&]() {
        hipLaunchKernelGGL(k_split_a, dim3(((size_t)M * K + 255) / 256), dim3(256), 0, 0, dA, dAp, M, K);
        hipLaunchKernelGGL(k_split_bt, dim3(N / 32, K / 32), dim3(256), 0, 0, dB, dBp, K, N);
        hipEventRecord(e1);
        hipLaunchKernelGGL(k_gemm, dim3(N / 64, M / 64), dim3(256), 0, 0, dAp, dBp, dC, M, K, N);
    };
    run(); hipDeviceSynchronize();
    hipEventRecord(e0); run(); hipEventRecord(e2); hipEventSynchronize(e2);
    float ms_all, ms_split; hipEventElapsedTime(&ms_all, e0, e2); hipEventElapsedTime(&ms_split, e0, e1);
    std::vector<uint64_t> hC((size_t)M * N);
    hipMemcpy(hC.data(), dC, hC.size() * 8, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int t = 0; t < 200; ++t) {
        int i = t == 0 ? 0 : rand() % M, j = t == 0 ? 0 : rand() % N;
        u128 s = 0;
        for (int kk = 0; kk < K; ++kk) s = (s + (u128)hA[(size_t)i * K + kk] * hB[(size_t)kk * N + j]) % P;
        if ((uint64_t)s != hC[(size_t)i * N + j]) ++bad;
    }
    printf("%dx%dx%d: %d/200 mismatches; split %.3f ms, gemm %.3f ms, total %.3f ms = %.2f TMAC/s\n", M, K, N, bad, ms_split,
           ms_all - ms_split, ms_all, (double)M * K * N / (ms_all * 1e-3) / 1e12);
    return 0;
}
