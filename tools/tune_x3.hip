// tune_x3.hip -- how fast do 12-byte-per-lane accesses (global_load/store_dwordx3) stream?
// Candidate storage format for 65..96-bit primes (12 instead of 16 bytes per element).
// Build: hipcc --offload-arch=gfx950 -O3 tools/tune_x3.hip -o /tmp/tune_x3
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef uint32_t u32x3 __attribute__((ext_vector_type(3)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
struct E96 { uint32_t x[3]; };

template <int U>
__global__ __launch_bounds__(256) void k3(const E96* a, const E96* b, E96* o, size_t n) {
    size_t base = (size_t)blockIdx.x * (256 * U) + threadIdx.x;
    u32x3 x[U], y[U];
#pragma unroll
    for (int q = 0; q < U; ++q) {
        size_t i = base + q * 256;
        if (i < n) {
            x[q] = __builtin_nontemporal_load((const u32x3*)&a[i]);
            y[q] = __builtin_nontemporal_load((const u32x3*)&b[i]);
        }
    }
#pragma unroll
    for (int q = 0; q < U; ++q) {
        size_t i = base + q * 256;
        if (i < n) {
            u32x3 r;
            r.x = x[q].x + y[q].x; r.y = x[q].y ^ y[q].y; r.z = x[q].z + y[q].z;
            __builtin_nontemporal_store(r, (u32x3*)&o[i]);
        }
    }
}
// 4 elements = 48 bytes = 3 dwordx4 per thread (lane stride 48 B)
__global__ __launch_bounds__(256) void k4(const u32x4* a, const u32x4* b, u32x4* o, size_t n4) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n4) {
        u32x4 x[3], y[3];
#pragma unroll
        for (int q = 0; q < 3; ++q) { x[q] = __builtin_nontemporal_load(a + 3 * i + q); y[q] = __builtin_nontemporal_load(b + 3 * i + q); }
#pragma unroll
        for (int q = 0; q < 3; ++q) __builtin_nontemporal_store(x[q] + y[q], o + 3 * i + q);
    }
}
// reference: 16 B per lane
__global__ __launch_bounds__(256) void k16(const u32x4* a, const u32x4* b, u32x4* o, size_t n) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) __builtin_nontemporal_store(__builtin_nontemporal_load(a + i) + __builtin_nontemporal_load(b + i), o + i);
}
int main() {
    const size_t n = 40000000;   // 480 MB per array
    void *a, *b, *o;
    hipMalloc(&a, n * 12 + 64); hipMalloc(&b, n * 12 + 64); hipMalloc(&o, n * 12 + 64);
    hipMemset(a, 1, n * 12); hipMemset(b, 2, n * 12);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](const char* name, auto launch) {
        for (int i = 0; i < 3; ++i) launch();
        hipEventRecord(e0);
        const int reps = 20;
        for (int i = 0; i < reps; ++i) launch();
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-28s %8.1f us  %7.0f GB/s\n", name, ms / reps * 1e3, 3.0 * n * 12 / (ms / reps * 1e-3) / 1e9);
        fflush(stdout);
    };
    run("dwordx3 x1", [&] { hipLaunchKernelGGL(k3<1>, dim3((n + 255) / 256), dim3(256), 0, 0, (const E96*)a, (const E96*)b, (E96*)o, n); });
    run("dwordx3 x2", [&] { hipLaunchKernelGGL(k3<2>, dim3((n + 511) / 512), dim3(256), 0, 0, (const E96*)a, (const E96*)b, (E96*)o, n); });
    run("dwordx3 x4", [&] { hipLaunchKernelGGL(k3<4>, dim3((n + 1023) / 1024), dim3(256), 0, 0, (const E96*)a, (const E96*)b, (E96*)o, n); });
    run("3 x dwordx4, stride 48", [&] { hipLaunchKernelGGL(k4, dim3((n / 4 + 255) / 256), dim3(256), 0, 0, (const u32x4*)a, (const u32x4*)b, (u32x4*)o, n / 4); });
    run("dwordx4 (16 B/lane)", [&] { hipLaunchKernelGGL(k16, dim3((n * 12 / 16 + 255) / 256), dim3(256), 0, 0, (const u32x4*)a, (const u32x4*)b, (u32x4*)o, n * 12 / 16); });
    return 0;
}
