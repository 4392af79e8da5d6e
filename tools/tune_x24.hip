// tune_x24.hip -- 24-byte elements (three-limb primes): per-lane 3 x dwordx2 at a 24-byte lane stride (what the library did
// until round 6) against wave-cooperative accesses -- every wave instruction moves 1 KiB of CONTIGUOUS memory (dwordx4 per
// lane) and the lanes' own elements (two per lane, 48 bytes) are sorted out through a per-wave LDS region.
// Build: hipcc --offload-arch=gfx950 -O3 tools/tune_x24.hip -o /tmp/tune_x24
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
struct E24 { uint64_t l[3]; };

__device__ __forceinline__ E24 op(const E24& x, const E24& y) {
    E24 r;
    r.l[0] = x.l[0] + y.l[0]; r.l[1] = x.l[1] ^ y.l[1]; r.l[2] = x.l[2] + 3 * y.l[2];
    return r;
}

// (a) one element per lane, 3 x dwordx2
__global__ __launch_bounds__(256) void k_lane(const E24* a, const E24* b, E24* o, size_t n) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const u32x2* pa = (const u32x2*)(a + i); const u32x2* pb = (const u32x2*)(b + i);
    u32x2 xa[3], xb[3];
#pragma unroll
    for (int q = 0; q < 3; ++q) { xa[q] = __builtin_nontemporal_load(pa + q); xb[q] = __builtin_nontemporal_load(pb + q); }
    E24 x, y;
#pragma unroll
    for (int q = 0; q < 3; ++q) { x.l[q] = xa[q].x | ((uint64_t)xa[q].y << 32); y.l[q] = xb[q].x | ((uint64_t)xb[q].y << 32); }
    E24 r = op(x, y);
    u32x2* po = (u32x2*)(o + i);
#pragma unroll
    for (int q = 0; q < 3; ++q) { u32x2 v; v.x = (uint32_t)r.l[q]; v.y = (uint32_t)(r.l[q] >> 32); __builtin_nontemporal_store(v, po + q); }
}

// (b) two elements per lane; a wave moves 3 KiB per array as 3 contiguous dwordx4 instructions; LDS sorts it out
template <int ARRS>   // LDS regions per wave: 1 = one region reused by a, b, out; 3 = one each
__global__ __launch_bounds__(256) void k_coop(const u32x4* a, const u32x4* b, u32x4* o, size_t npair) {
    __shared__ u32x4 buf[4][ARRS][192];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const size_t wbase = ((size_t)blockIdx.x * 4 + wv) * 64;          // first pair of this wave
    if (wbase >= npair) return;
    const bool whole = wbase + 64 <= npair;
    if (!whole) {      // ragged last wave: per lane
        size_t i = wbase + lane;
        if (i < npair) {
            const E24* ea = (const E24*)a + 2 * i; const E24* eb = (const E24*)b + 2 * i; E24* eo = (E24*)o + 2 * i;
            eo[0] = op(ea[0], eb[0]); eo[1] = op(ea[1], eb[1]);
        }
        return;
    }
    const u32x4* pa = a + wbase * 3; const u32x4* pb = b + wbase * 3; u32x4* po = o + wbase * 3;
    u32x4 ra[3], rb[3];
#pragma unroll
    for (int q = 0; q < 3; ++q) { ra[q] = __builtin_nontemporal_load(pa + q * 64 + lane); rb[q] = __builtin_nontemporal_load(pb + q * 64 + lane); }
    u32x4* la = buf[wv][0]; u32x4* lb = buf[wv][ARRS > 1 ? 1 : 0]; u32x4* lo = buf[wv][ARRS > 2 ? 2 : 0];
    u32x4 ta[3], tb[3];
#pragma unroll
    for (int q = 0; q < 3; ++q) la[q * 64 + lane] = ra[q];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int q = 0; q < 3; ++q) ta[q] = la[lane * 3 + q];
    if (ARRS == 1) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); }
#pragma unroll
    for (int q = 0; q < 3; ++q) lb[q * 64 + lane] = rb[q];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int q = 0; q < 3; ++q) tb[q] = lb[lane * 3 + q];
    // registers -> two elements
    uint32_t wa[12] = {ta[0].x, ta[0].y, ta[0].z, ta[0].w, ta[1].x, ta[1].y, ta[1].z, ta[1].w, ta[2].x, ta[2].y, ta[2].z, ta[2].w};
    uint32_t wb[12] = {tb[0].x, tb[0].y, tb[0].z, tb[0].w, tb[1].x, tb[1].y, tb[1].z, tb[1].w, tb[2].x, tb[2].y, tb[2].z, tb[2].w};
    uint32_t wo[12];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        E24 x, y;
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            x.l[q] = wa[6 * e + 2 * q] | ((uint64_t)wa[6 * e + 2 * q + 1] << 32);
            y.l[q] = wb[6 * e + 2 * q] | ((uint64_t)wb[6 * e + 2 * q + 1] << 32);
        }
        E24 r = op(x, y);
#pragma unroll
        for (int q = 0; q < 3; ++q) { wo[6 * e + 2 * q] = (uint32_t)r.l[q]; wo[6 * e + 2 * q + 1] = (uint32_t)(r.l[q] >> 32); }
    }
    if (ARRS < 3) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); }
#pragma unroll
    for (int q = 0; q < 3; ++q) { u32x4 v; v.x = wo[4 * q]; v.y = wo[4 * q + 1]; v.z = wo[4 * q + 2]; v.w = wo[4 * q + 3]; lo[lane * 3 + q] = v; }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int q = 0; q < 3; ++q) __builtin_nontemporal_store(lo[q * 64 + lane], po + q * 64 + lane);
}

// reference: 16 B per lane, contiguous (what a 16-byte-element field does)
__global__ __launch_bounds__(256) void k16(const u32x4* a, const u32x4* b, u32x4* o, size_t n) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) __builtin_nontemporal_store(__builtin_nontemporal_load(a + i) + __builtin_nontemporal_load(b + i), o + i);
}

int main() {
    const size_t n = 10000000;   // 240 MB per array
    const size_t bytes = n * 24;
    void *a, *b, *o, *o2;
    hipMalloc(&a, bytes + 64); hipMalloc(&b, bytes + 64); hipMalloc(&o, bytes + 64); hipMalloc(&o2, bytes + 64);
    uint64_t* ha = (uint64_t*)malloc(bytes); uint64_t* hb = (uint64_t*)malloc(bytes);
    uint64_t s = 88172645463325252ull;
    for (size_t i = 0; i < 3 * n; ++i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; ha[i] = s; s ^= s << 13; s ^= s >> 7; s ^= s << 17; hb[i] = s; }
    hipMemcpy(a, ha, bytes, hipMemcpyHostToDevice); hipMemcpy(b, hb, bytes, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](const char* name, auto launch) {
        for (int i = 0; i < 3; ++i) launch();
        hipEventRecord(e0);
        const int reps = 20;
        for (int i = 0; i < reps; ++i) launch();
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-44s %8.1f us  %7.0f GB/s\n", name, ms / reps * 1e3, 3.0 * bytes / (ms / reps * 1e-3) / 1e9);
        fflush(stdout);
    };
    run("per lane: 3 x dwordx2, stride 24", [&] { hipLaunchKernelGGL(k_lane, dim3((n + 255) / 256), dim3(256), 0, 0, (const E24*)a, (const E24*)b, (E24*)o, n); });
    const size_t np = n / 2;
    run("wave-cooperative, 1 LDS region", [&] { hipLaunchKernelGGL(k_coop<1>, dim3((np + 255) / 256), dim3(256), 0, 0, (const u32x4*)a, (const u32x4*)b, (u32x4*)o2, np); });
    run("wave-cooperative, 3 LDS regions", [&] { hipLaunchKernelGGL(k_coop<3>, dim3((np + 255) / 256), dim3(256), 0, 0, (const u32x4*)a, (const u32x4*)b, (u32x4*)o2, np); });
    run("dwordx4 contiguous (16 B/lane)", [&] { hipLaunchKernelGGL(k16, dim3((bytes / 16 + 255) / 256), dim3(256), 0, 0, (const u32x4*)a, (const u32x4*)b, (u32x4*)o, bytes / 16); });
    // correctness of the cooperative kernel against the per-lane one
    hipLaunchKernelGGL(k_lane, dim3((n + 255) / 256), dim3(256), 0, 0, (const E24*)a, (const E24*)b, (E24*)o, n);
    hipLaunchKernelGGL(k_coop<1>, dim3((np + 255) / 256), dim3(256), 0, 0, (const u32x4*)a, (const u32x4*)b, (u32x4*)o2, np);
    hipDeviceSynchronize();
    uint64_t* h1 = (uint64_t*)malloc(bytes); uint64_t* h2 = (uint64_t*)malloc(bytes);
    hipMemcpy(h1, o, bytes, hipMemcpyDeviceToHost); hipMemcpy(h2, o2, bytes, hipMemcpyDeviceToHost);
    size_t bad = 0;
    for (size_t i = 0; i < 3 * n; ++i) bad += h1[i] != h2[i];
    printf("mismatches (1 region): %zu\n", bad);
    hipLaunchKernelGGL(k_coop<3>, dim3((np + 255) / 256), dim3(256), 0, 0, (const u32x4*)a, (const u32x4*)b, (u32x4*)o2, np);
    hipDeviceSynchronize();
    hipMemcpy(h2, o2, bytes, hipMemcpyDeviceToHost);
    bad = 0;
    for (size_t i = 0; i < 3 * n; ++i) bad += h1[i] != h2[i];
    printf("mismatches (3 regions): %zu\n", bad);
    return 0;
}
