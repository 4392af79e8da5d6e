// tools/tune_stream.hip -- standalone tuning probe (not part of the library).
// Sweeps launch geometry / unroll / cache policy for the streaming kernels and measures the
// integer-ALU ceiling of each field policy, so that kernels.hpp can be set from measurements.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/tune_stream.hip -o build/tune_stream
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "../mpyc_amd/csrc/policy_build.hpp"

using namespace ffgpu;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

struct alignas(16) P2 { uint64_t w[2]; };

template <bool NT> __device__ __forceinline__ P2 ldp(const P2* p) {
    if constexpr (NT) {
        P2 r;
        r.w[0] = __builtin_nontemporal_load(&p->w[0]);
        r.w[1] = __builtin_nontemporal_load(&p->w[1]);
        return r;
    } else {
        return *p;
    }
}
template <bool NT> __device__ __forceinline__ void stp(P2* p, const P2& v) {
    if constexpr (NT) {
        __builtin_nontemporal_store(v.w[0], &p->w[0]);
        __builtin_nontemporal_store(v.w[1], &p->w[1]);
    } else {
        *p = v;
    }
}

// R reads, W writes per pack; compute = mulmod of the first two reads (if R>=2)
template <class F, int R, int W, int U, bool NTL, bool NTS, int BS, bool CHUNK>
__global__ __launch_bounds__(BS) void k_stream(F f, const P2* __restrict__ in, size_t in_stride, P2* __restrict__ out,
                                               size_t out_stride, size_t nvec) {
    size_t gsz = (size_t)gridDim.x * BS;
    size_t start, end, step;
    if constexpr (CHUNK) {
        size_t per = (nvec + gridDim.x - 1) / gridDim.x;
        start = (size_t)blockIdx.x * per + threadIdx.x;
        end = (size_t)(blockIdx.x + 1) * per;
        if (end > nvec) end = nvec;
        step = BS;
    } else {
        start = (size_t)blockIdx.x * BS + threadIdx.x;
        end = nvec;
        step = gsz;
    }
    for (size_t i = start; i < end; i += step * U) {
        P2 x[U][R > 0 ? R : 1];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            size_t j = i + (size_t)u * step;
            if (j < end) {
#pragma unroll
                for (int r = 0; r < R; ++r) x[u][r] = ldp<NTL>(in + (size_t)r * in_stride + j);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            size_t j = i + (size_t)u * step;
            if (j < end) {
                P2 y;
                if constexpr (R >= 2) {
                    y.w[0] = f.mul(x[u][0].w[0], x[u][1].w[0]);
                    y.w[1] = f.mul(x[u][0].w[1], x[u][1].w[1]);
#pragma unroll
                    for (int r = 2; r < R; ++r) { y.w[0] = f.add(y.w[0], x[u][r].w[0]); y.w[1] = f.add(y.w[1], x[u][r].w[1]); }
                } else if constexpr (R == 1) {
                    y = x[u][0];
                } else {
                    y.w[0] = j; y.w[1] = ~j;
                }
                if constexpr (W == 0) {
                    if (y.w[0] == 0x123456789abcdefull && y.w[1] == 77) out[j] = y;  // keep loads alive
                }
#pragma unroll
                for (int w = 0; w < W; ++w) {
                    P2 z = y;
                    z.w[0] += w;   // distinct data per row, negligible ALU
                    stp<NTS>(out + (size_t)w * out_stride + j, z);
                }
            }
        }
    }
}

struct Bufs {
    std::vector<P2*> in, out;
    size_t stride;
};

template <class K>
static float time_it(K launch, int sets, int reps) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int s = 0; s < sets; ++s) launch(s);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    for (int r = 0; r < reps; ++r) for (int s = 0; s < sets; ++s) launch(s);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
    return ms / (reps * sets);
}

typedef PM64<false, true> M61;

template <int R, int W, int U, bool NTL, bool NTS, int BS, bool CHUNK>
static void run_cfg(const M61& f, Bufs& b, size_t nvec, int bpc, int cus, const char* tag) {
    size_t want = (nvec + (size_t)BS * U - 1) / ((size_t)BS * U);
    size_t cap = bpc > 0 ? (size_t)bpc * cus : want;
    unsigned grid = (unsigned)(want < cap ? want : cap);
    int sets = (int)b.in.size();
    float ms = time_it([&](int s) {
        hipLaunchKernelGGL((k_stream<M61, R, W, U, NTL, NTS, BS, CHUNK>), dim3(grid), dim3(BS), 0, 0, f, b.in[s], b.stride,
                           b.out[s], b.stride, nvec);
    }, sets, 5);
    double bytes = (double)(R + W) * 16.0 * nvec;
    printf("%-10s R%dW%d U%d ntl%d nts%d bs%-4d %s bpc%-3d grid%-6u  %8.2f us  %7.1f GB/s\n", tag, R, W, U, (int)NTL, (int)NTS, BS,
           CHUNK ? "chunk " : "stride", bpc, grid, ms * 1e3, bytes / (ms * 1e-3) / 1e9);
}

// ---- ALU ceiling: chained field multiplications in registers -------------------
template <class F, int CH>
__global__ __launch_bounds__(256) void k_alu(F f, typename F::word seed, typename F::word* out, int iters) {
    typename F::word x[CH];
    typename F::word y = seed;
#pragma unroll
    for (int c = 0; c < CH; ++c) { x[c] = seed; if constexpr (sizeof(typename F::word) == 16) x[c].lo += threadIdx.x + c; else x[c] += threadIdx.x + c; }
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int c = 0; c < CH; ++c) x[c] = f.mul(x[c], y);
    }
    typename F::word acc = x[0];
#pragma unroll
    for (int c = 1; c < CH; ++c) acc = f.add(acc, x[c]);
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = acc;
}

template <class F>
static void alu_probe(const char* name, const PolicyBlob& pb, int cus) {
    F f; memcpy(&f, pb.bytes, sizeof(F));
    typename F::word* out;
    unsigned grid = cus * 8;
    CK(hipMalloc(&out, (size_t)grid * 256 * sizeof(typename F::word)));
    typename F::word seed; memset(&seed, 0x5a, sizeof(seed));
    seed = f.reduce_raw(seed);
    const int CH = 4, iters = 2000;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k_alu<F, CH>), dim3(grid), dim3(256), 0, 0, f, seed, out, 10);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL((k_alu<F, CH>), dim3(grid), dim3(256), 0, 0, f, seed, out, iters);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    double muls = (double)grid * 256 * CH * iters * F::EPW;
    printf("ALU %-18s %8.2f G mul/s  (%.3f ms)\n", name, muls / (ms * 1e-3) / 1e9, ms);
    CK(hipFree(out));
}

int main(int argc, char** argv) {
    size_t n = 10000000;
    int sets = 6;
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    int cus = prop.multiProcessorCount;
    printf("device %s CUs %d clock %d MHz\n", prop.name, cus, prop.clockRate / 1000);
    size_t nvec = n / 2;
    Bufs b; b.stride = (nvec + 15) / 16 * 16;
    for (int s = 0; s < sets; ++s) {
        P2 *i_, *o_;
        CK(hipMalloc(&i_, 3 * b.stride * sizeof(P2)));
        CK(hipMalloc(&o_, 7 * b.stride * sizeof(P2)));
        CK(hipMemset(i_, 0x11 + s, 3 * b.stride * sizeof(P2)));
        CK(hipMemset(o_, 0, 7 * b.stride * sizeof(P2)));
        b.in.push_back(i_); b.out.push_back(o_);
    }
    PolicyBlob pb; build_prime_policy(&pb, ((ff_u128)1 << 61) - 1);
    M61 f; memcpy(&f, pb.bytes, sizeof(f));

    // ceilings
    run_cfg<1, 1, 2, false, false, 256, false>(f, b, nvec, 8, cus, "copy");
    run_cfg<1, 1, 2, true, true, 256, false>(f, b, nvec, 8, cus, "copy");
    run_cfg<1, 1, 4, false, false, 256, false>(f, b, nvec, 8, cus, "copy");
    run_cfg<1, 1, 4, false, false, 256, false>(f, b, nvec, 0, cus, "copy");
    run_cfg<1, 0, 4, false, false, 256, false>(f, b, nvec, 8, cus, "read");
    run_cfg<1, 0, 4, true, false, 256, false>(f, b, nvec, 8, cus, "read");
    run_cfg<0, 1, 4, false, false, 256, false>(f, b, nvec, 8, cus, "write");
    run_cfg<0, 1, 4, false, true, 256, false>(f, b, nvec, 8, cus, "write");

    // mul kernel 2R+1W: geometry sweep
    int bpcs[] = {2, 4, 8, 16, 0};
    for (int bpc : bpcs) {
        run_cfg<2, 1, 1, false, false, 256, false>(f, b, nvec, bpc, cus, "mul");
        run_cfg<2, 1, 2, false, false, 256, false>(f, b, nvec, bpc, cus, "mul");
        run_cfg<2, 1, 4, false, false, 256, false>(f, b, nvec, bpc, cus, "mul");
    }
    run_cfg<2, 1, 2, true, false, 256, false>(f, b, nvec, 8, cus, "mul");
    run_cfg<2, 1, 2, false, true, 256, false>(f, b, nvec, 8, cus, "mul");
    run_cfg<2, 1, 2, true, true, 256, false>(f, b, nvec, 8, cus, "mul");
    run_cfg<2, 1, 4, true, true, 256, false>(f, b, nvec, 8, cus, "mul");
    run_cfg<2, 1, 2, true, true, 256, false>(f, b, nvec, 0, cus, "mul");
    run_cfg<2, 1, 2, false, false, 512, false>(f, b, nvec, 4, cus, "mul");
    run_cfg<2, 1, 2, false, false, 1024, false>(f, b, nvec, 2, cus, "mul");
    run_cfg<2, 1, 2, false, false, 128, false>(f, b, nvec, 16, cus, "mul");
    run_cfg<2, 1, 2, false, false, 64, false>(f, b, nvec, 32, cus, "mul");
    run_cfg<2, 1, 2, false, false, 256, true>(f, b, nvec, 8, cus, "mul");
    run_cfg<2, 1, 4, false, false, 256, true>(f, b, nvec, 8, cus, "mul");
    run_cfg<2, 1, 2, false, false, 256, true>(f, b, nvec, 4, cus, "mul");
    run_cfg<2, 1, 2, true, true, 256, true>(f, b, nvec, 8, cus, "mul");

    // split-like 2R+3W and 4R+7W, recombine-like 3R+1W, 7R+1W
    for (int bpc : {4, 8, 0}) {
        run_cfg<2, 3, 1, false, false, 256, false>(f, b, nvec, bpc, cus, "split13");
        run_cfg<2, 3, 2, false, false, 256, false>(f, b, nvec, bpc, cus, "split13");
        run_cfg<2, 3, 1, true, true, 256, false>(f, b, nvec, bpc, cus, "split13");
        run_cfg<3, 1, 1, false, false, 256, false>(f, b, nvec, bpc, cus, "rec3");
        run_cfg<3, 1, 2, false, false, 256, false>(f, b, nvec, bpc, cus, "rec3");
        run_cfg<3, 1, 1, true, true, 256, false>(f, b, nvec, bpc, cus, "rec3");
    }
    // (4R+7W and 7R+1W use the 3-row input / 7-row output buffers: reads wrap is avoided by R<=3)
    run_cfg<3, 7, 1, false, false, 256, false>(f, b, nvec, 8, cus, "split37~");
    run_cfg<3, 7, 1, true, true, 256, false>(f, b, nvec, 8, cus, "split37~");
    run_cfg<3, 7, 1, false, false, 256, false>(f, b, nvec, 0, cus, "split37~");
    run_cfg<3, 7, 1, false, false, 256, false>(f, b, nvec, 4, cus, "split37~");

    // bs / nt with uncapped grid
    run_cfg<2, 1, 1, true, true, 256, false>(f, b, nvec, 0, cus, "mul");
    run_cfg<2, 1, 1, true, true, 512, false>(f, b, nvec, 0, cus, "mul");
    run_cfg<2, 1, 1, true, true, 1024, false>(f, b, nvec, 0, cus, "mul");
    run_cfg<2, 1, 1, true, true, 128, false>(f, b, nvec, 0, cus, "mul");
    run_cfg<2, 1, 1, true, false, 256, false>(f, b, nvec, 0, cus, "mul");
    run_cfg<2, 1, 1, false, true, 256, false>(f, b, nvec, 0, cus, "mul");
    run_cfg<1, 1, 1, true, true, 256, false>(f, b, nvec, 0, cus, "copy");
    run_cfg<1, 1, 1, false, false, 256, false>(f, b, nvec, 0, cus, "copy");
    // how much is fixed (ramp + tail + launch gap) vs steady state: same kernel, smaller/larger n
    for (size_t frac = 8; frac >= 1; frac /= 2) {
        char tag[32]; snprintf(tag, sizeof(tag), "mul/n%zu", frac);
        run_cfg<2, 1, 1, true, true, 256, false>(f, b, nvec / frac, 0, cus, tag);
    }
    // ALU ceilings
    {
        PolicyBlob q;
        build_prime_policy(&q, ((ff_u128)1 << 61) - 1); alu_probe<PM64<false, true> >("PM64 2^61-1", q, cus);
        build_prime_policy(&q, (ff_u128)0 - 189 + ((ff_u128)0)); // placeholder, replaced below
        build_prime_policy(&q, (((ff_u128)1 << 64) - 189)); alu_probe<PM64<true, false> >("PM64 2^64-189", q, cus);
        build_prime_policy(&q, (((ff_u128)1 << 40) - 87)); if (q.kind == POL_PM64_GEN) alu_probe<PM64<false, false> >("PM64 2^40-87", q, cus);
        build_prime_policy(&q, (ff_u128)6616326157076047771ull); alu_probe<RC64>("RC64 generic63", q, cus);
        build_prime_policy(&q, (ff_u128)2147483647u); alu_probe<RC32>("RC32 2^31-1", q, cus);
        build_prime_policy(&q, (ff_u128)0 - 173); alu_probe<PM128<true> >("PM128 2^128-173", q, cus);
        build_prime_policy(&q, (((ff_u128)1 << 127) - 1)); alu_probe<PM128<false> >("PM128 2^127-1", q, cus);
        ff_u128 g = ((ff_u128)0xC2B2AE3D27D4EB4Full << 64) | 0x165667B19E377A5Full;
        build_prime_policy(&q, g); if (q.kind == POL_MONT128) alu_probe<MONT128>("MONT128 generic", q, cus);
        uint64_t m8[3] = {0x11b, 0, 0}; build_binary_policy(&q, m8, 3); alu_probe<GF2P8>("GF2P8 (x4 packed)", q, cus);
        uint64_t m64[3] = {0x1b, 1, 0}; build_binary_policy(&q, m64, 3); alu_probe<GF2W64>("GF2W64 n=64", q, cus);
        uint64_t m128[3] = {0x87, 0, 1}; build_binary_policy(&q, m128, 3); alu_probe<GF2W128>("GF2W128 n=128", q, cus);
    }
    return 0;
}
