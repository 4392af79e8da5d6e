// tools/tune_split.hip -- probe for the many-stream kernels (share generation writes m rows):
// which launch shape / row stride / cache policy keeps HBM efficient with 4 read + 7 write streams.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <bool NT> __device__ __forceinline__ u32x4 ld(const u32x4* p) { if constexpr (NT) return __builtin_nontemporal_load(p); else return *p; }
template <bool NT> __device__ __forceinline__ void st(u32x4* p, u32x4 v) { if constexpr (NT) __builtin_nontemporal_store(v, p); else *p = v; }

// R read rows, W write rows, U packs per thread; CONTIG: the U packs of a block are adjacent 4 KiB chunks
template <int R, int W, int U, bool NTL, bool NTS, bool CONTIG>
__global__ __launch_bounds__(256) void k(const u32x4* __restrict__ in, size_t is, u32x4* __restrict__ out, size_t os, size_t nvec) {
    size_t gsz = (size_t)gridDim.x * 256;
    size_t base = CONTIG ? (size_t)blockIdx.x * 256 * U + threadIdx.x : (size_t)blockIdx.x * 256 + threadIdx.x;
    size_t step = CONTIG ? 256 : gsz;
    u32x4 x[U][R];
#pragma unroll
    for (int u = 0; u < U; ++u) { size_t j = base + u * step; if (j < nvec) {
#pragma unroll
        for (int r = 0; r < R; ++r) x[u][r] = ld<NTL>(in + r * is + j); } }
#pragma unroll
    for (int u = 0; u < U; ++u) { size_t j = base + u * step; if (j < nvec) {
        u32x4 y = x[u][0];
#pragma unroll
        for (int r = 1; r < R; ++r) y = y * 3u + x[u][r];
#pragma unroll
        for (int w = 0; w < W; ++w) { y = y * 5u + (uint32_t)w; st<NTS>(out + w * os + j, y); } } }
}

template <int R, int W, int U, bool NTL, bool NTS, bool CONTIG>
static void run(std::vector<u32x4*>& in, std::vector<u32x4*>& out, size_t nvec, size_t is, size_t os, const char* tag) {
    unsigned grid = (unsigned)((nvec + 256 * U - 1) / (256 * U));
    int sets = (int)in.size(), reps = 5;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int s = 0; s < sets; ++s) hipLaunchKernelGGL((k<R, W, U, NTL, NTS, CONTIG>), dim3(grid), dim3(256), 0, 0, in[s], is, out[s], os, nvec);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    for (int r = 0; r < reps; ++r) for (int s = 0; s < sets; ++s) hipLaunchKernelGGL((k<R, W, U, NTL, NTS, CONTIG>), dim3(grid), dim3(256), 0, 0, in[s], is, out[s], os, nvec);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps * sets;
    printf("%-8s R%dW%d U%d ntl%d nts%d %s is=%zu os=%zu grid%-6u %8.2f us %7.1f GB/s\n", tag, R, W, U, (int)NTL, (int)NTS, CONTIG ? "contig" : "spread", is, os, grid, ms * 1e3,
           (double)(R + W) * 16 * nvec / (ms * 1e-3) / 1e9);
}

int main() {
    setvbuf(stdout, NULL, _IONBF, 0);
    size_t nvec = 5000000;      // 10^7 64-bit elements
    size_t maxs = nvec + 4096;
    std::vector<u32x4*> in, out;
    for (int s = 0; s < 4; ++s) { u32x4 *a, *b; CK(hipMalloc(&a, 7 * maxs * 16)); CK(hipMalloc(&b, 7 * maxs * 16)); CK(hipMemset(a, s + 1, 7 * maxs * 16)); CK(hipMemset(b, 0, 7 * maxs * 16)); in.push_back(a); out.push_back(b); }
    size_t s0 = nvec;   // 80,000,000 B rows
    run<4, 7, 1, true, true, false>(in, out, nvec, s0, s0, "base");
    run<4, 7, 1, false, false, false>(in, out, nvec, s0, s0, "base");
    run<4, 7, 1, true, false, false>(in, out, nvec, s0, s0, "base");
    run<4, 7, 1, false, true, false>(in, out, nvec, s0, s0, "base");
    run<4, 7, 2, true, true, true>(in, out, nvec, s0, s0, "contig");
    run<4, 7, 4, true, true, true>(in, out, nvec, s0, s0, "contig");
    run<4, 7, 2, true, true, false>(in, out, nvec, s0, s0, "spread");
    size_t strides[] = {nvec + 16, nvec + 64, nvec + 256, nvec + 272, nvec + 1040, nvec + 4096};
    for (size_t st_ : strides) run<4, 7, 1, true, true, false>(in, out, nvec, st_, st_, "stride");
    // power-of-two row pitch (worst case for channel aliasing): 4 Mi packs = 64 MiB rows, n = 4Mi packs
    run<4, 7, 1, true, true, false>(in, out, 4194304, 4194304, 4194304, "pow2");
    run<4, 7, 1, true, true, false>(in, out, 4194304, 4194304 + 272, 4194304 + 272, "pow2+");
    // fewer streams for reference
    run<2, 3, 1, true, true, false>(in, out, nvec, s0, s0, "ref");
    run<4, 1, 1, true, true, false>(in, out, nvec, s0, s0, "ref");
    run<1, 7, 1, true, true, false>(in, out, nvec, s0, s0, "ref");
    run<1, 3, 1, true, true, false>(in, out, nvec, s0, s0, "ref");
    run<1, 1, 1, true, true, false>(in, out, nvec, s0, s0, "ref");
    run<7, 1, 1, true, true, false>(in, out, nvec, s0, s0, "ref");
    return 0;
}
