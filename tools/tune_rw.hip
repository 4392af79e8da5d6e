// tune_rw.hip -- read-only, write-only and mixed streaming rates (16 B per lane, non-temporal, uncapped grid):
// what is the ceiling a 7-reads-1-write kernel (recombination) or a 1-read-3-writes kernel (share generation)
// can hope for?   hipcc --offload-arch=gfx950 -O3 tools/tune_rw.hip -o build/tune_rw
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
template <int R, int Wn>
__global__ __launch_bounds__(256) void k(const u32x4* __restrict__ in, u32x4* __restrict__ out, size_t n, size_t stride) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    u32x4 acc = {0, 0, 0, 0};
#pragma unroll
    for (int r = 0; r < R; ++r) acc += __builtin_nontemporal_load(in + r * stride + i);
    if (Wn == 0) {
        if (acc.x == 0x12345678u && acc.y == 0x9abcdef0u) out[i] = acc;     // never true: keeps the loads alive
    } else {
#pragma unroll
        for (int w = 0; w < Wn; ++w) __builtin_nontemporal_store(acc + (uint32_t)w, out + w * stride + i);
    }
}
int main() {
    const size_t n = 5000000;           // 80 MB per row
    const size_t stride = n + 17 * 16;  // skewed pitch as the library does
    u32x4 *in, *out;
    hipMalloc(&in, 8 * stride * 16); hipMalloc(&out, 8 * stride * 16);
    hipMemset(in, 1, 8 * stride * 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](const char* name, int r, int w, auto kern) {
        for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, dim3((n + 255) / 256), dim3(256), 0, 0, in, out, n, stride);
        hipEventRecord(e0);
        const int reps = 20;
        for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kern, dim3((n + 255) / 256), dim3(256), 0, 0, in, out, n, stride);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-22s %7.1f us  %6.0f GB/s\n", name, ms / reps * 1e3, (double)(r + w) * n * 16 / (ms / reps * 1e-3) / 1e9);
        fflush(stdout);
    };
    run("read 1 row", 1, 0, k<1, 0>);
    run("read 4 rows", 4, 0, k<4, 0>);
    run("read 8 rows", 8, 0, k<8, 0>);
    run("1 read 1 write", 1, 1, k<1, 1>);
    run("2 reads 1 write", 2, 1, k<2, 1>);
    run("3 reads 1 write", 3, 1, k<3, 1>);
    run("7 reads 1 write", 7, 1, k<7, 1>);
    run("1 read 3 writes", 1, 3, k<1, 3>);
    run("3 reads 3 writes", 3, 3, k<3, 3>);
    run("1 read 7 writes", 1, 7, k<1, 7>);
    return 0;
}
