// api.hip -- the extern "C" boundary of libffgpu.so (declared in include/ffgpu.h).
// Host-side only: classifies the modulus, builds the field policy, dispatches
// to the per-policy launcher tables (ops_*.hip).  No arithmetic on array data
// happens on the host: if the GPU path cannot run, the call returns an error.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <mutex>

#include "../../include/ffgpu.h"
#include "kernels.hpp"
#include "policy_build.hpp"

using namespace ffgpu;

// per-policy launcher tables (one translation unit each, see ops_*.hip)
const FieldOps* ffgpu_ops_pm64_mersenne();   // PM64<false,true>
const FieldOps* ffgpu_ops_pm64_k64();        // PM64<true,false>
const FieldOps* ffgpu_ops_pm64_gen();        // PM64<false,false>
const FieldOps* ffgpu_ops_rc64();
const FieldOps* ffgpu_ops_rc32();
const FieldOps* ffgpu_ops_pm128_k128();      // PM128<true>
const FieldOps* ffgpu_ops_pm128_gen();       // PM128<false>
const FieldOps* ffgpu_ops_pm96();            // PM96
const FieldOps* ffgpu_ops_pm192();           // PM192
const FieldOps* ffgpu_ops_mont192();         // MONT192
const FieldOps* ffgpu_ops_mont128();
const FieldOps* ffgpu_ops_gf2p8();
const FieldOps* ffgpu_ops_gf2w32();
const FieldOps* ffgpu_ops_gf2w64();
const FieldOps* ffgpu_ops_gf2w128();
int ffgpu_sbox_build_lut(const void* gf2p8_policy, const uint8_t* rows8, uint8_t b, uint8_t* lut256);
int ffgpu_launch_sbox(const uint8_t* lut256, int device, const void* in, void* out, size_t n, hipStream_t st);
int ffgpu_launch_gf8_to_bits(int device, const void* in, const void* addend, void* out, size_t n, hipStream_t st);
int ffgpu_launch_gf8_mask_open(const void* policy, int device, const void* const* rows, const uint64_t* coef2, int nrows,
                               const void* const* rbits, const uint64_t* mu2, int np, void* out, size_t n, hipStream_t st);
int ffgpu_launch_gf8_bits_affine_fold(const void* policy, int device, const uint64_t* m2, const uint64_t* bias2, const void* c,
                                      const void* rbits, size_t ybr, void* out, size_t ybo, size_t n, int nbatch,
                                      hipStream_t st);
int ffgpu_launch_gf8_group8(const void* policy, int device, const uint64_t* m2, const uint64_t* bias2, int fold,
                            const void* in, void* out, size_t ngroups, hipStream_t st);
int ffgpu_launch_copy(int device, const void* src, void* dst, size_t bytes, hipStream_t st);
int ffgpu_launch_valu_probe(int device, int op, int iters, int waves_per_simd, void* scratch32, double* out, hipStream_t st);
int ffgpu_gf8_build_tables(const void* policy, void* tables_out);
int ffgpu_gf2w_build_rtable(const void* policy, int limbs, void* rtable_out);
int ffgpu_launch_gf2w_recombine(const void* policy, int limbs, int device, const void* const* rows, const uint64_t* lam2,
                                int k, void* out, size_t n, hipStream_t st);
int ffgpu_launch_gf8_sbox_layer(const void* policy, int device, const void* x, size_t xs, const void* r, size_t rs, void* out,
                                size_t os, const void* tables_dev, const uint64_t* lam2, const uint64_t* mu2, int t, int m,
                                size_t n, hipStream_t st, const ffgpu::RngArgs* rng);
void ffgpu_gf8_sbox_layer_tables(const void* policy, const void* mul_tables, const uint64_t* m2, const uint64_t* bias2,
                                 unsigned char* out);
int ffgpu_launch_gf2w_mul_win(const void* policy, int limbs, const void* rtable, int device, const void* a,
                              const void* b, void* out, size_t n, hipStream_t st);
size_t ffgpu_launch_gf2w64_mul_bitsliced(const void* policy, int device, const void* a, const void* b, void* out, size_t n,
                                         hipStream_t st);
int ffgpu_launch_gf8_mul_tab(const void* tables, int device, const void* a, const void* b, void* out, size_t n,
                             hipStream_t st);

struct ffgpu_ctx {
    int kind;
    int reduction;
    int device;
    int elem_bytes;
    int policy_kind;
    const FieldOps* ops;
    uint64_t rng_r[2];  // 2^W mod p for the keystream sampler
    ffgpu::Tuning tune; // run-time switches, read from the environment when the context is created
    int gf8_tab_min;    // GF(2^n<=8): arrays of at least this many elements multiply through LDS tables
    alignas(16) unsigned char gf8_tables[1536];
    int gf2w_limbs;     // GF(2^n), 9 <= n <= 128: 1 or 2 limbs -> windowed multiplication kernel
    alignas(16) unsigned char gf2w_rtable[256];
    // last S-box table built for this context (depends only on rows8, b)
    uint8_t sbox_key[9];
    int sbox_valid;
    uint8_t sbox_lut[256];
    alignas(16) unsigned char policy[128];
    uint64_t modulus[3];
    // grow-only device scratch of ffgpu_matmul (digit planes, split-K slabs), ONE BUFFER PER STREAM: launches on
    // different streams never share scratch, and a buffer is only freed after its own stream has drained
    struct Scratch {
        hipStream_t stream;
        void* ptr;
        size_t bytes;
        int used;
    } scratch[8];
    std::mutex* scratch_mu;
    void* gf8_tables_dev;   // device copy of gf8_tables (lazily, for the fused GF(2^n<=8) product)
    // tables of the fused S-box layer (ffgpu_gf256_sbox_layer) for the last affine map used with this context
    void* sbl_tables_dev;
    unsigned char sbl_key[72];
    int sbl_valid;
    // opt-in timing of the most recent compute call (ffgpu_ctx_set_timing / ffgpu_last_kernel_ms)
    int timing, timed;
    hipEvent_t ev0, ev1;
    // timing mode 2 (ffgpu_busy_ms): one event pair per compute call, harvested into acc_ms
    enum { ACC_MAX = 256 };
    hipEvent_t acc_ev[2 * ACC_MAX];
    int acc_made, acc_n;
    double acc_ms;
    unsigned long long acc_calls;
};

// sum the elapsed time of the recorded event pairs of accumulate mode into acc_ms (waits for the last one)
static void acc_harvest(ffgpu_ctx* c) {
    if (c->acc_n == 0) return;
    (void)hipEventSynchronize(c->acc_ev[2 * (c->acc_n - 1) + 1]);
    for (int i = 0; i < c->acc_n; ++i) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, c->acc_ev[2 * i], c->acc_ev[2 * i + 1]) == hipSuccess) c->acc_ms += ms;
    }
    c->acc_calls += (unsigned long long)c->acc_n;
    c->acc_n = 0;
}

static thread_local char g_hip_err[256] = "";

static int hip_fail(hipError_t e, const char* what) {
    snprintf(g_hip_err, sizeof(g_hip_err), "%s: %s", what, hipGetErrorString(e));
    return FFGPU_EHIP;
}
#define HIPCHK(call)                                     \
    do {                                                 \
        hipError_t e_ = (call);                          \
        if (e_ != hipSuccess) return hip_fail(e_, #call); \
    } while (0)

static int launch_status(int rc) {
    if (rc == 0) return FFGPU_OK;
    if (rc & 0x10000) return hip_fail((hipError_t)(rc & 0xffff), "kernel launch");
    if (rc == 2) return FFGPU_ENOTSUP;
    return FFGPU_EINVAL;
}

namespace ffgpu {
LaunchCfg launch_cfg(int device) {
    static std::mutex mu;
    static LaunchCfg cache[64];
    static bool have[64];
    std::lock_guard<std::mutex> g(mu);
    int d = (device >= 0 && device < 64) ? device : 0;
    if (!have[d]) {
        int cus = 256;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, d) != hipSuccess || cus <= 0)
            cus = 256;
        int bpc = 0;  // uncapped
        const char* e = getenv("FFGPU_BLOCKS_PER_CU");
        if (e && atoi(e) >= 0) bpc = atoi(e);
        cache[d].num_cu = cus;
        cache[d].blocks_per_cu = bpc;
        have[d] = true;
    }
    return cache[d];
}
}  // namespace ffgpu

struct DeviceGuard {
    int prev;
    bool switched;
    explicit DeviceGuard(int dev) : prev(-1), switched(false) {
        if (hipGetDevice(&prev) == hipSuccess && prev != dev) {
            if (hipSetDevice(dev) == hipSuccess) switched = true;
        }
    }
    ~DeviceGuard() {
        if (switched) (void)hipSetDevice(prev);
    }
};

// brackets the launches of one API call with two events when the context has timing switched on
struct LaunchTimer {
    ffgpu_ctx* c;
    hipStream_t st;
    int slot;
    LaunchTimer(ffgpu_ctx* ctx, hipStream_t s) : c(ctx && ctx->timing ? ctx : nullptr), st(s), slot(-1) {
        if (!c) return;
        if (c->timing == 2) {
            if (c->acc_n == ffgpu_ctx::ACC_MAX) acc_harvest(c);
            if (c->acc_n == c->acc_made) {
                if (hipEventCreate(&c->acc_ev[2 * c->acc_made]) != hipSuccess ||
                    hipEventCreate(&c->acc_ev[2 * c->acc_made + 1]) != hipSuccess) {
                    c = nullptr;
                    return;
                }
                ++c->acc_made;
            }
            slot = c->acc_n++;
            (void)hipEventRecord(c->acc_ev[2 * slot], st);
        } else {
            (void)hipEventRecord(c->ev0, st);
        }
    }
    ~LaunchTimer() {
        if (!c) return;
        if (slot >= 0) {
            (void)hipEventRecord(c->acc_ev[2 * slot + 1], st);
        } else {
            (void)hipEventRecord(c->ev1, st);
            c->timed = 1;
        }
    }
};

static const FieldOps* ops_for(int kind) {
    switch (kind) {
        case POL_PM64_MERSENNE: return ffgpu_ops_pm64_mersenne();
        case POL_PM64_K64: return ffgpu_ops_pm64_k64();
        case POL_PM64_GEN: return ffgpu_ops_pm64_gen();
        case POL_RC64: return ffgpu_ops_rc64();
        case POL_RC32: return ffgpu_ops_rc32();
        case POL_PM128_K128: return ffgpu_ops_pm128_k128();
        case POL_PM128_GEN: return ffgpu_ops_pm128_gen();
        case POL_PM96: return ffgpu_ops_pm96();
        case POL_PM192: return ffgpu_ops_pm192();
        case POL_MONT192: return ffgpu_ops_mont192();
        case POL_MONT128: return ffgpu_ops_mont128();
        case POL_GF2P8: return ffgpu_ops_gf2p8();
        case POL_GF2W32: return ffgpu_ops_gf2w32();
        case POL_GF2W64: return ffgpu_ops_gf2w64();
        case POL_GF2W128: return ffgpu_ops_gf2w128();
        default: return nullptr;
    }
}

extern "C" {

int ffgpu_abi_version(void) { return FFGPU_ABI_VERSION; }

const char* ffgpu_strerror(int status) {
    switch (status) {
        case FFGPU_OK: return "ok";
        case FFGPU_EINVAL: return "invalid argument";
        case FFGPU_ENOTSUP: return "field or size not supported by this build";
        case FFGPU_EHIP: return "HIP runtime error";
        case FFGPU_ESTALE: return "interprocess mapping does not show the exported row (stale mapping)";
        case FFGPU_EMODULUS: return "unusable modulus";
        case FFGPU_ENOMEM: return "out of device memory";
        default: return "unknown status";
    }
}

const char* ffgpu_last_hip_error(void) { return g_hip_err; }

int ffgpu_device_count(int* count) {
    if (!count) return FFGPU_EINVAL;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        *count = 0;
        return hip_fail(e, "hipGetDeviceCount");
    }
    *count = n;
    return FFGPU_OK;
}

int ffgpu_device_pci_bus_id(int device, char* buf, int len) {
    if (!buf || len < 16 || device < 0) return FFGPU_EINVAL;
    buf[0] = 0;
    hipError_t e = hipDeviceGetPCIBusId(buf, len, device);
    if (e != hipSuccess) return hip_fail(e, "hipDeviceGetPCIBusId");
    return FFGPU_OK;
}

int ffgpu_ctx_create(int kind, const uint64_t* modulus, int nlimbs, int device, ffgpu_ctx** out) {
    if (!modulus || !out || nlimbs < 1 || nlimbs > 3 || device < 0) return FFGPU_EINVAL;
    ffgpu_ctx* c = (ffgpu_ctx*)calloc(1, sizeof(ffgpu_ctx));
    if (!c) return FFGPU_ENOMEM;
    c->scratch_mu = new std::mutex();
    c->kind = kind;
    c->device = device;
    for (int i = 0; i < nlimbs; ++i) c->modulus[i] = modulus[i];
    PolicyBlob pb;
    memset(&pb, 0, sizeof(pb));
    int rc;
    if (kind == FFGPU_PRIME) {
        rc = build_prime_policy3(&pb, modulus, nlimbs);
    } else if (kind == FFGPU_BINARY) {
        rc = build_binary_policy(&pb, modulus, nlimbs);
    } else {
        rc = FFGPU_EINVAL;
    }
    if (rc == FFGPU_OK) {
        c->ops = ops_for(pb.kind);
        if (!c->ops) rc = FFGPU_ENOTSUP;
    }
    if (rc != FFGPU_OK) {
        free(c);
        return rc;
    }
    memcpy(c->policy, pb.bytes, sizeof(c->policy));
    c->reduction = pb.reduction;
    c->elem_bytes = pb.elem_bytes;
    c->policy_kind = pb.kind;
    rng_const(pb, c->rng_r);
    {   // the library's switches: read here, once per context -- no call path looks at the environment
        const char* e = getenv("FFGPU_MM_MFMA");
        c->tune.mm_mfma = e ? (atoi(e) != 0) : 1;
        e = getenv("FFGPU_MM_MFMA_MIN");
        c->tune.mm_mfma_min = e ? atof(e) : 8e7;
        e = getenv("FFGPU_GF2W_BITSLICED");
        c->tune.gf2w_bitsliced = e ? (atoi(e) != 0) : 1;
    }
    c->gf2w_limbs = 0;
    if (pb.kind == POL_GF2W64 || pb.kind == POL_GF2W128) {
        // sparse moduli (all MPyC defaults) multiply in registers through the integer multiplier
        // (fields.hpp ff_clmul*); dense moduli use the 4-bit window kernel with LDS tables
        bool in_regs;
        if (pb.kind == POL_GF2W128) {
            GF2W128 f;
            memcpy(&f, pb.bytes, sizeof(f));
            in_regs = (f.fast & 1) != 0;
        } else {
            GF2W64 f;
            memcpy(&f, pb.bytes, sizeof(f));
            in_regs = f.n <= 32 || (f.fast & 1) != 0;
        }
        if (!in_regs) {
            c->gf2w_limbs = pb.kind == POL_GF2W128 ? 2 : 1;
            ffgpu_gf2w_build_rtable(c->policy, c->gf2w_limbs, c->gf2w_rtable);
        }
    }
    c->gf8_tab_min = 0;
    if (pb.kind == POL_GF2P8 && ffgpu_gf8_build_tables(c->policy, c->gf8_tables) == 0) {
        c->gf8_tab_min = 1 << 18;   // below this the shift-xor kernel has lower latency
    }
    *out = c;
    return FFGPU_OK;
}

// device copy of the GF(2^n<=8) log/antilog tables for the fused share-generation kernel (made on first use:
// contexts can be created and classified on machines without a GPU)
static const void* gf8_tables_on_device(ffgpu_ctx* ctx) {
    static std::mutex mu;
    if (!ctx->gf8_tab_min) return nullptr;                 // not a GF(2^n<=8) context with tables
    std::lock_guard<std::mutex> g(mu);
    if (!ctx->gf8_tables_dev) {
        void* d = nullptr;
        if (hipMalloc(&d, sizeof(ctx->gf8_tables)) != hipSuccess) return nullptr;
        if (hipMemcpy(d, ctx->gf8_tables, sizeof(ctx->gf8_tables), hipMemcpyHostToDevice) != hipSuccess) {
            (void)hipFree(d);
            return nullptr;
        }
        ctx->gf8_tables_dev = d;
    }
    return ctx->gf8_tables_dev;
}

int ffgpu_ctx_destroy(ffgpu_ctx* ctx) {
    if (ctx) {
        DeviceGuard g(ctx->device);
        for (auto& sc : ctx->scratch)
            if (sc.ptr) (void)hipFree(sc.ptr);
        delete ctx->scratch_mu;
        ctx->scratch_mu = nullptr;
    }
    if (ctx && ctx->gf8_tables_dev) {
        DeviceGuard g(ctx->device);
        (void)hipFree(ctx->gf8_tables_dev);
    }
    if (ctx && ctx->sbl_tables_dev) {
        DeviceGuard g(ctx->device);
        (void)hipFree(ctx->sbl_tables_dev);
    }
    if (ctx && ctx->acc_made) {
        DeviceGuard g(ctx->device);
        for (int i = 0; i < 2 * ctx->acc_made; ++i) (void)hipEventDestroy(ctx->acc_ev[i]);
    }
    if (ctx && ctx->ev0) {
        DeviceGuard g(ctx->device);
        (void)hipEventDestroy(ctx->ev0);
        (void)hipEventDestroy(ctx->ev1);
    }
    free(ctx);
    return FFGPU_OK;
}
int ffgpu_ctx_set_timing(ffgpu_ctx* ctx, int enable) {
    if (!ctx) return FFGPU_EINVAL;
    if (enable && !ctx->ev0) {
        DeviceGuard g(ctx->device);
        HIPCHK(hipEventCreate(&ctx->ev0));
        HIPCHK(hipEventCreate(&ctx->ev1));
    }
    if (ctx->timing == 2 && enable != 2) {
        DeviceGuard g(ctx->device);
        acc_harvest(ctx);
    }
    ctx->timing = enable == 2 ? 2 : (enable ? 1 : 0);
    ctx->timed = 0;
    return FFGPU_OK;
}
int ffgpu_busy_ms(ffgpu_ctx* ctx, double* ms, unsigned long long* calls, int reset) {
    if (!ctx) return FFGPU_EINVAL;
    DeviceGuard g(ctx->device);
    acc_harvest(ctx);
    if (ms) *ms = ctx->acc_ms;
    if (calls) *calls = ctx->acc_calls;
    if (reset) {
        ctx->acc_ms = 0.0;
        ctx->acc_calls = 0;
    }
    return FFGPU_OK;
}
int ffgpu_last_kernel_ms(ffgpu_ctx* ctx, float* ms) {
    if (!ctx || !ms || !ctx->timed) return FFGPU_EINVAL;
    DeviceGuard g(ctx->device);
    HIPCHK(hipEventSynchronize(ctx->ev1));
    HIPCHK(hipEventElapsedTime(ms, ctx->ev0, ctx->ev1));
    return FFGPU_OK;
}
int ffgpu_ctx_elem_bytes(const ffgpu_ctx* ctx) { return ctx ? ctx->elem_bytes : -1; }
int ffgpu_ctx_reduction(const ffgpu_ctx* ctx) { return ctx ? ctx->reduction : -1; }
int ffgpu_ctx_device(const ffgpu_ctx* ctx) { return ctx ? ctx->device : -1; }

int ffgpu_malloc(ffgpu_ctx* ctx, size_t bytes, void** dptr) {
    if (!ctx || !dptr) return FFGPU_EINVAL;
    DeviceGuard g(ctx->device);
    hipError_t e = hipMalloc(dptr, bytes ? bytes : 16);
    if (e == hipErrorOutOfMemory) return FFGPU_ENOMEM;
    if (e != hipSuccess) return hip_fail(e, "hipMalloc");
    return FFGPU_OK;
}
int ffgpu_free(ffgpu_ctx* ctx, void* dptr) {
    if (!ctx) return FFGPU_EINVAL;
    DeviceGuard g(ctx->device);
    HIPCHK(hipFree(dptr));
    return FFGPU_OK;
}
int ffgpu_h2d(ffgpu_ctx* ctx, void* dst, const void* host_src, size_t bytes, void* stream) {
    if (!ctx || (bytes && (!dst || !host_src))) return FFGPU_EINVAL;
    DeviceGuard g(ctx->device);
    HIPCHK(hipMemcpyAsync(dst, host_src, bytes, hipMemcpyHostToDevice, (hipStream_t)stream));
    return FFGPU_OK;
}
int ffgpu_d2h(ffgpu_ctx* ctx, void* host_dst, const void* src, size_t bytes, void* stream) {
    if (!ctx || (bytes && (!host_dst || !src))) return FFGPU_EINVAL;
    DeviceGuard g(ctx->device);
    HIPCHK(hipMemcpyAsync(host_dst, src, bytes, hipMemcpyDeviceToHost, (hipStream_t)stream));
    return FFGPU_OK;
}
// ---- device buffers across co-located party processes (include/ffgpu.h "device-side wire") ----
// first and last 16 bytes of a row in one small synchronous copy (2 rows of 16 bytes, pitch = bytes - 16)
static int ipc_canary(const void* row, size_t bytes, unsigned char* out32) {
    if (bytes < 32) return FFGPU_EINVAL;
    HIPCHK(hipMemcpy2D(out32, 16, row, bytes - 16, 16, 2, hipMemcpyDeviceToHost));
    return FFGPU_OK;
}
int ffgpu_ipc_export(ffgpu_ctx* ctx, const void* ptr, size_t bytes, unsigned char* handle, unsigned long long* offset,
                     unsigned char* canary32, void* stream) {
    if (!ctx || !ptr || !handle || !offset) return FFGPU_EINVAL;
    static_assert(sizeof(hipIpcMemHandle_t) == FFGPU_IPC_HANDLE_BYTES, "handle size");
    DeviceGuard g(ctx->device);
    HIPCHK(hipStreamSynchronize((hipStream_t)stream));          // the row is complete before anybody can open it
    hipDeviceptr_t base = nullptr;
    size_t size = 0;
    HIPCHK(hipMemGetAddressRange(&base, &size, (hipDeviceptr_t)ptr));
    hipIpcMemHandle_t h;
    HIPCHK(hipIpcGetMemHandle(&h, base));
    memcpy(handle, &h, sizeof(h));
    *offset = (unsigned long long)((const char*)ptr - (const char*)base);
    if (canary32) return ipc_canary(ptr, bytes, canary32);
    return FFGPU_OK;
}
int ffgpu_ipc_open(ffgpu_ctx* ctx, const unsigned char* handle, void** base) {
    if (!ctx || !handle || !base) return FFGPU_EINVAL;
    DeviceGuard g(ctx->device);
    hipIpcMemHandle_t h;
    memcpy(&h, handle, sizeof(h));
    void* p = nullptr;
    HIPCHK(hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess));
    *base = p;
    return FFGPU_OK;
}
int ffgpu_ipc_read(ffgpu_ctx* ctx, const void* base, unsigned long long offset, void* dst, size_t bytes,
                   const unsigned char* expect_canary32, void* stream) {
    if (!ctx || (bytes && (!base || !dst))) return FFGPU_EINVAL;
    DeviceGuard g(ctx->device);
    const char* src = (const char*)base + offset;
    if (expect_canary32) {
        unsigned char got[32];
        int rc = ipc_canary(src, bytes, got);
        if (rc != FFGPU_OK) return rc;
        if (memcmp(got, expect_canary32, 32) != 0) return FFGPU_ESTALE;
    }
    HIPCHK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    HIPCHK(hipStreamSynchronize((hipStream_t)stream));          // the copy HAS run when the caller acknowledges the row
    return FFGPU_OK;
}
int ffgpu_ipc_read_reduced(ffgpu_ctx* ctx, const void* base, unsigned long long offset, void* dst, size_t n,
                           const unsigned char* expect_canary32, void* stream) {
    if (!ctx || (n && (!base || !dst))) return FFGPU_EINVAL;
    const char* src = (const char*)base + offset;
    if (expect_canary32) {
        DeviceGuard g(ctx->device);
        unsigned char got[32];
        int rc = ipc_canary(src, n * (size_t)ctx->elem_bytes, got);
        if (rc != FFGPU_OK) return rc;
        if (memcmp(got, expect_canary32, 32) != 0) return FFGPU_ESTALE;
    }
    int rc = ffgpu_reduce(ctx, src, dst, n, stream);            // ONE pass: the peer's row is read through the mapping
    if (rc != FFGPU_OK) return rc;
    DeviceGuard g(ctx->device);
    HIPCHK(hipStreamSynchronize((hipStream_t)stream));
    return FFGPU_OK;
}
int ffgpu_ipc_close(ffgpu_ctx* ctx, void* base) {
    if (!ctx || !base) return FFGPU_EINVAL;
    DeviceGuard g(ctx->device);
    HIPCHK(hipIpcCloseMemHandle(base));
    return FFGPU_OK;
}

int ffgpu_stream_sync(ffgpu_ctx* ctx, void* stream) {
    if (!ctx) return FFGPU_EINVAL;
    DeviceGuard g(ctx->device);
    HIPCHK(hipStreamSynchronize((hipStream_t)stream));
    return FFGPU_OK;
}

#define ARGCHK(cond) \
    do {             \
        if (!(cond)) return FFGPU_EINVAL; \
    } while (0)

static int do_ew2(ffgpu_ctx* ctx, int op, const void* a, const void* b, void* out, size_t n, void* stream) {
    ARGCHK(ctx);
    if (n == 0) return FFGPU_OK;
    ARGCHK(a && b && out);
    DeviceGuard g(ctx->device);
    LaunchTimer lt(ctx, (hipStream_t)stream);
    return launch_status(ctx->ops->ew2(ctx->policy, ctx->device, op, a, b, out, n, (hipStream_t)stream));
}
static int do_ew1(ffgpu_ctx* ctx, int op, const void* a, const uint64_t* s, void* out, size_t n, void* stream) {
    ARGCHK(ctx);
    if (n == 0) return FFGPU_OK;
    ARGCHK(a && out);
    DeviceGuard g(ctx->device);
    LaunchTimer lt(ctx, (hipStream_t)stream);
    return launch_status(ctx->ops->ew1(ctx->policy, ctx->device, op, a, s, out, n, (hipStream_t)stream));
}

int ffgpu_reduce(ffgpu_ctx* ctx, const void* raw, void* out, size_t n, void* stream) {
    return do_ew1(ctx, OP_REDUCE, raw, nullptr, out, n, stream);
}
int ffgpu_add(ffgpu_ctx* ctx, const void* a, const void* b, void* out, size_t n, void* stream) {
    return do_ew2(ctx, OP_ADD, a, b, out, n, stream);
}
int ffgpu_sub(ffgpu_ctx* ctx, const void* a, const void* b, void* out, size_t n, void* stream) {
    return do_ew2(ctx, OP_SUB, a, b, out, n, stream);
}
int ffgpu_mul(ffgpu_ctx* ctx, const void* a, const void* b, void* out, size_t n, void* stream) {
    if (ctx && ctx->gf8_tab_min && n >= (size_t)ctx->gf8_tab_min && a && b && out) {
        DeviceGuard g(ctx->device);
        LaunchTimer lt(ctx, (hipStream_t)stream);
        return launch_status(ffgpu_launch_gf8_mul_tab(ctx->gf8_tables, ctx->device, a, b, out, n,
                                                      (hipStream_t)stream));
    }
    if (ctx && ctx->gf2w_limbs && n && a && b && out) {
        DeviceGuard g(ctx->device);
        LaunchTimer lt(ctx, (hipStream_t)stream);
        return launch_status(ffgpu_launch_gf2w_mul_win(ctx->policy, ctx->gf2w_limbs, ctx->gf2w_rtable, ctx->device,
                                                       a, b, out, n, (hipStream_t)stream));
    }
    if (ctx && ctx->policy_kind == POL_GF2W64 && ctx->tune.gf2w_bitsliced && a && b && out && n >= ((size_t)1 << 21)) {
        // GF(2^64) with the default modulus: bit-sliced product for all whole pairs of elements, the element-wise kernel for
        // the odd last one -- both launches under ONE timer scope (ffgpu_last_kernel_ms reports the call, not its tail)
        DeviceGuard g(ctx->device);
        LaunchTimer lt(ctx, (hipStream_t)stream);
        const size_t done = ffgpu_launch_gf2w64_mul_bitsliced(ctx->policy, ctx->device, a, b, out, n, (hipStream_t)stream);
        if (done && hipGetLastError() != hipSuccess) return FFGPU_EHIP;
        if (done == n) return FFGPU_OK;
        if (done)
            return launch_status(ctx->ops->ew2(ctx->policy, ctx->device, OP_MUL, (const char*)a + 8 * done, (const char*)b + 8 * done,
                                               (char*)out + 8 * done, n - done, (hipStream_t)stream));
    }
    return do_ew2(ctx, OP_MUL, a, b, out, n, stream);
}
int ffgpu_neg(ffgpu_ctx* ctx, const void* a, void* out, size_t n, void* stream) {
    return do_ew1(ctx, OP_NEG, a, nullptr, out, n, stream);
}
int ffgpu_add_scalar(ffgpu_ctx* ctx, const void* a, const uint64_t* s, void* out, size_t n, void* stream) {
    ARGCHK(s);
    return do_ew1(ctx, OP_ADD, a, s, out, n, stream);
}
int ffgpu_mul_scalar(ffgpu_ctx* ctx, const void* a, const uint64_t* s, void* out, size_t n, void* stream) {
    ARGCHK(s);
    return do_ew1(ctx, OP_MUL, a, s, out, n, stream);
}
int ffgpu_rsub_scalar(ffgpu_ctx* ctx, const void* a, const uint64_t* s, void* out, size_t n, void* stream) {
    ARGCHK(s);
    return do_ew1(ctx, OP_RSUB, a, s, out, n, stream);
}
int ffgpu_muladd(ffgpu_ctx* ctx, const void* a, const void* b, const void* c, void* out, size_t n,
                 void* stream) {
    ARGCHK(ctx);
    if (n == 0) return FFGPU_OK;
    ARGCHK(a && b && c && out);
    DeviceGuard g(ctx->device);
    LaunchTimer lt(ctx, (hipStream_t)stream);
    return launch_status(ctx->ops->muladd(ctx->policy, ctx->device, a, b, c, out, n, (hipStream_t)stream));
}

static int make_exp(const uint64_t* e, int limbs, ExpArgs* ex) {
    if (!e || limbs < 1 || limbs > 3) return FFGPU_EINVAL;
    ex->post = 0;
    ex->e[0] = e[0];
    ex->e[1] = limbs > 1 ? e[1] : 0;
    ex->e[2] = limbs > 2 ? e[2] : 0;
    int nb = 0;
    for (int i = 191; i >= 0; --i)
        if ((ex->e[i >> 6] >> (i & 63)) & 1) {
            nb = i + 1;
            break;
        }
    ex->nbits = nb;
    return FFGPU_OK;
}

// products (squarings + multiplications) of ff_pow_chain<true> for this exponent: the same decisions, on the host
static int exp_chain_cost(const ExpArgs& ex) {
    auto bit = [&](int i) -> int { return (int)((ex.e[i >> 6] >> (i & 63)) & 1u); };
    if (ex.nbits <= 1) return 0;
    int run = 0;
    for (int i = ex.nbits - 1; i >= 0 && bit(i); --i) ++run;
    int cost = 0, i = ex.nbits - 2;
    if (run >= 12) {
        int have = 1;
        for (int b = 30 - __builtin_clz((unsigned)run); b >= 0; --b) {
            cost += have + 1;
            have *= 2;
            if ((run >> b) & 1) {
                cost += 2;
                ++have;
            }
        }
        i = ex.nbits - 1 - run;
    }
    int ones = 0;
    for (int b = i; b >= 0; --b) ones += bit(b);
    if (i < 4 || ones <= 8 + (i + 1) / 8) return cost + (i + 1) + ones;
    cost += 8;                                              // a^2 and the seven odd powers
    while (i >= 0) {
        if (!bit(i)) {
            ++cost;
            --i;
            continue;
        }
        int j = i - 3 > 0 ? i - 3 : 0;
        while (!bit(j)) ++j;
        cost += (i - j + 1) + 1;
        i = j - 1;
    }
    return cost;
}
// e = 3 e' + 1 with a shorter chain for e' (+ 3 products for r^3 * a): hand over (e', post = 1)
static void exp_try_cube_form(ExpArgs* ex) {
    if (ex->nbits < 16) return;
    uint64_t q[3];
    unsigned __int128 rem = 0;
    for (int l = 2; l >= 0; --l) {
        const unsigned __int128 cur = (rem << 64) | ex->e[l];
        q[l] = (uint64_t)(cur / 3);
        rem = cur % 3;
    }
    if (rem != 1) return;
    ExpArgs alt;
    if (make_exp(q, 3, &alt) != FFGPU_OK || alt.nbits == 0) return;
    if (exp_chain_cost(alt) + 3 < exp_chain_cost(*ex)) {
        alt.post = 1;
        *ex = alt;
    }
}

int ffgpu_pow(ffgpu_ctx* ctx, const void* a, const uint64_t* host_exp, int exp_limbs, void* out, size_t n,
              void* stream) {
    ARGCHK(ctx);
    ExpArgs ex;
    int rc = make_exp(host_exp, exp_limbs, &ex);
    if (rc != FFGPU_OK) return rc;
    exp_try_cube_form(&ex);
    if (n == 0) return FFGPU_OK;
    ARGCHK(a && out);
    DeviceGuard g(ctx->device);
    LaunchTimer lt(ctx, (hipStream_t)stream);
    if (ex.nbits == 0) {
        // a^0 = 1 (also for a = 0, as pow(0, 0, p) = 1): 0*a + 1
        uint64_t zero[3] = {0, 0, 0}, one[3] = {1, 0, 0};
        rc = launch_status(ctx->ops->ew1(ctx->policy, ctx->device, OP_MUL, a, zero, out, n, (hipStream_t)stream));
        if (rc != FFGPU_OK) return rc;
        return launch_status(ctx->ops->ew1(ctx->policy, ctx->device, OP_ADD, out, one, out, n, (hipStream_t)stream));
    }
    return launch_status(ctx->ops->pow(ctx->policy, ctx->device, a, &ex, out, n, (hipStream_t)stream));
}

// three-limb primes: exponents derived from p by limb arithmetic
static void sub_small3(const uint64_t p[3], uint64_t d, uint64_t out[3]) {     // p - d, d small, p >= d
    out[0] = p[0] - d;
    const uint64_t b = p[0] < d;
    out[1] = p[1] - b;
    out[2] = p[2] - ((p[1] < b) ? 1 : 0);
}
static void shr1_3(uint64_t x[3]) {
    x[0] = (x[0] >> 1) | (x[1] << 63);
    x[1] = (x[1] >> 1) | (x[2] << 63);
    x[2] >>= 1;
}
static bool three_limb_prime(const ffgpu_ctx* ctx) { return ctx->kind == FFGPU_PRIME && ctx->modulus[2] != 0; }

// exponent q - 2 for x^-1 = x^(q-2); false for the two-element fields, where x^-1 = x
static bool inverse_exponent(const ffgpu_ctx* ctx, ExpArgs* ex) {
    if (three_limb_prime(ctx)) {
        uint64_t e3[3];
        sub_small3(ctx->modulus, 2, e3);
        make_exp(e3, 3, ex);
        return true;
    }
    ff_u128 q;
    if (ctx->kind == FFGPU_PRIME) {
        q = ff_make128(ctx->modulus[1], ctx->modulus[0]);
    } else {
        int deg = ctx->modulus[2] ? 128 : (ctx->modulus[1] ? 64 + (63 - __builtin_clzll(ctx->modulus[1]))
                                                           : 63 - __builtin_clzll(ctx->modulus[0]));
        q = deg == 128 ? (ff_u128)0 : ((ff_u128)1 << deg);
    }
    ff_u128 e = q - 2;
    uint64_t el[2] = {ff_lo(e), ff_hi(e)};
    make_exp(el, 2, ex);
    if (ex->nbits == 0) {
        ex->e[0] = 1;
        ex->nbits = 1;
        return false;
    }
    return true;
}

int ffgpu_sqrt_cl(ffgpu_ctx* ctx, const void* a, void* out, size_t n, void* stream) {
    ARGCHK(ctx);
    if (ctx->kind != FFGPU_PRIME || (ctx->modulus[0] & 3) != 1) return FFGPU_ENOTSUP;
    if (n == 0) return FFGPU_OK;
    ARGCHK(a && out);
    ExpArgs eleg, elad;
    if (three_limb_prime(ctx)) {
        uint64_t l1[3], l2[3];
        sub_small3(ctx->modulus, 1, l1);                   // (p-1)/2
        shr1_3(l1);
        sub_small3(ctx->modulus, 1, l2);                   // (p+1)/2 = (p-1)/2 + 1 (p odd: the low limb cannot wrap)
        shr1_3(l2);
        l2[0] += 1;
        if (l2[0] == 0 && ++l2[1] == 0) ++l2[2];
        make_exp(l1, 3, &eleg);
        make_exp(l2, 3, &elad);
    } else {
        ff_u128 p = ff_make128(ctx->modulus[1], ctx->modulus[0]);
        ff_u128 e1 = (p - 1) >> 1, e2 = (p >> 1) + 1;      // (p-1)/2 and (p+1)/2 (p odd; no overflow at 128 bits)
        uint64_t l1[2] = {ff_lo(e1), ff_hi(e1)}, l2[2] = {ff_lo(e2), ff_hi(e2)};
        make_exp(l1, 2, &eleg);
        make_exp(l2, 2, &elad);
    }
    DeviceGuard g(ctx->device);
    LaunchTimer lt(ctx, (hipStream_t)stream);
    return launch_status(ctx->ops->sqrt_cl(ctx->policy, ctx->device, a, &eleg, &elad, out, n, (hipStream_t)stream));
}

int ffgpu_gauss(ffgpu_ctx* ctx, void* a, int n, int ncols, size_t batch, int mode, void* det_out,
                void* dev_singular, void* stream) {
    ARGCHK(ctx && n >= 0 && ncols >= n && (mode == 0 || mode == 1));
    if (batch == 0) return FFGPU_OK;
    ARGCHK(dev_singular && (mode == 0 || det_out));
    if (n == 0) return FFGPU_OK;
    ARGCHK(a);
    DeviceGuard g(ctx->device);
    LaunchTimer lt(ctx, (hipStream_t)stream);
    ExpArgs ex;
    inverse_exponent(ctx, &ex);
    if (hipMemsetAsync(dev_singular, 0, batch * sizeof(int), (hipStream_t)stream) != hipSuccess) return FFGPU_EHIP;
    return launch_status(ctx->ops->gauss(ctx->policy, ctx->device, a, n, ncols, batch, mode, &ex,
                                         mode ? det_out : nullptr, (int*)dev_singular, (hipStream_t)stream));
}

int ffgpu_inv(ffgpu_ctx* ctx, const void* a, void* out, size_t n, void* dev_zero_flag, void* stream) {
    ARGCHK(ctx);
    if (n == 0) return FFGPU_OK;
    ARGCHK(a && out);
    if (three_limb_prime(ctx)) {
        ExpArgs ex3;
        inverse_exponent(ctx, &ex3);
        DeviceGuard g3(ctx->device);
        LaunchTimer lt3(ctx, (hipStream_t)stream);
        return launch_status(ctx->ops->inv(ctx->policy, ctx->device, a, &ex3, out, n, (int*)dev_zero_flag,
                                           (hipStream_t)stream));
    }
    // exponent q - 2 (order of the multiplicative group minus one)
    ff_u128 q;
    if (ctx->kind == FFGPU_PRIME) {
        q = ff_make128(ctx->modulus[1], ctx->modulus[0]);
    } else {
        int deg = ctx->modulus[2] ? 128 : (ctx->modulus[1] ? 64 + (63 - __builtin_clzll(ctx->modulus[1]))
                                                           : 63 - __builtin_clzll(ctx->modulus[0]));
        q = deg == 128 ? (ff_u128)0 : ((ff_u128)1 << deg);   // 2^128 wraps to 0; q-2 below is still right
    }
    ff_u128 e = q - 2;
    uint64_t el[2] = {ff_lo(e), ff_hi(e)};
    DeviceGuard g(ctx->device);
    LaunchTimer lt(ctx, (hipStream_t)stream);
    if (ctx->kind == FFGPU_PRIME && q == 2) {   // GF(2): 1^-1 = 1
        ExpArgs one;
        one.e[0] = 1; one.e[1] = one.e[2] = 0; one.nbits = 1; one.post = 0;
        return launch_status(ctx->ops->inv(ctx->policy, ctx->device, a, &one, out, n, (int*)dev_zero_flag,
                                           (hipStream_t)stream));
    }
    ExpArgs ex;
    int rc = make_exp(el, 2, &ex);
    if (rc != FFGPU_OK) return rc;
    if (ex.nbits == 0) { ex.e[0] = 1; ex.nbits = 1; }        // GF(2^1): q - 2 = 0 -> x^-1 = x
    return launch_status(ctx->ops->inv(ctx->policy, ctx->device, a, &ex, out, n, (int*)dev_zero_flag,
                                       (hipStream_t)stream));
}

int ffgpu_beaver_combine(ffgpu_ctx* ctx, const void* z, const void* x, const void* y, const void* d, const void* e,
                         int add_de, void* out, size_t n, void* stream) {
    ARGCHK(ctx);
    if (n == 0) return FFGPU_OK;
    ARGCHK(z && x && y && d && e && out);
    DeviceGuard g(ctx->device);
    LaunchTimer lt(ctx, (hipStream_t)stream);
    return launch_status(ctx->ops->beaver(ctx->policy, ctx->device, z, x, y, d, e, out, add_de ? 1 : 0, n,
                                          (hipStream_t)stream));
}

static int do_split(ffgpu_ctx* ctx, const void* a, const void* b, bool fused, const void* coeffs,
                    size_t coeff_stride, int t, int m, void* shares, size_t share_stride, size_t n,
                    void* stream) {
    ARGCHK(ctx);
    ARGCHK(m >= 1 && t >= 0 && t < m);  // thresha.py:26 "0 <= t < m"
    if (n == 0) return FFGPU_OK;
    ARGCHK(a && shares && (!fused || b) && (t == 0 || coeffs));
    ARGCHK((m == 1 || share_stride >= n) && (t <= 1 || coeff_stride >= n));
    DeviceGuard g(ctx->device);
    LaunchTimer lt(ctx, (hipStream_t)stream);
    return launch_status(ctx->ops->split(ctx->policy, ctx->device, a, fused ? b : nullptr, coeffs,
                                         coeff_stride, t, m, shares, share_stride, n, (hipStream_t)stream,
                                         nullptr));
}

static int make_rng(const ffgpu_ctx* ctx, const uint8_t* key32, uint64_t nonce, int rounds, RngArgs* ra) {
    if (!key32) return FFGPU_EINVAL;
    if (rounds == 0) rounds = 20;
    if (rounds != 20 && rounds != 12 && rounds != 8) return FFGPU_EINVAL;
    memset(ra, 0, sizeof(*ra));
    memcpy(ra->rk.key, key32, 32);  // little-endian words, as RFC 8439
    ra->rk.nonce[0] = (uint32_t)nonce;
    ra->rk.nonce[1] = (uint32_t)(nonce >> 32);
    ra->rk.rounds = (uint32_t)rounds;
    ra->r0 = ctx->rng_r[0];
    ra->r1 = ctx->rng_r[1];
    return FFGPU_OK;
}

static int do_split_rng(ffgpu_ctx* ctx, const void* a, const void* b, bool fused, const uint8_t* key32,
                        uint64_t nonce, int rounds, int t, int m, void* shares, size_t share_stride, size_t n,
                        void* stream) {
    ARGCHK(ctx);
    ARGCHK(m >= 1 && t >= 0 && t < m);
    RngArgs ra;
    int rc = make_rng(ctx, key32, nonce, rounds, &ra);
    if (rc != FFGPU_OK) return rc;
    if (n == 0) return FFGPU_OK;
    ARGCHK(a && shares && (!fused || b));
    ARGCHK(m == 1 || share_stride >= n);
    DeviceGuard g(ctx->device);
    LaunchTimer lt(ctx, (hipStream_t)stream);
    if (fused && t > 0) ra.aux = gf8_tables_on_device(ctx);
    return launch_status(ctx->ops->split(ctx->policy, ctx->device, a, fused ? b : nullptr, nullptr, 0, t, m,
                                         shares, share_stride, n, (hipStream_t)stream, t > 0 ? &ra : nullptr));
}
int ffgpu_ctx_scalar_limbs(const ffgpu_ctx* ctx) { return (ctx && ctx->elem_bytes == 24) ? 3 : 2; }

size_t ffgpu_rng_state_bytes(void) { return sizeof(RngKey); }

int ffgpu_rng_state_init(ffgpu_ctx* ctx, void* dev_state, const uint8_t* host_key32, uint64_t nonce, int rounds,
                         void* stream) {
    ARGCHK(ctx && dev_state);
    RngArgs ra;
    int rc = make_rng(ctx, host_key32, nonce, rounds, &ra);
    if (rc != FFGPU_OK) return rc;
    DeviceGuard g(ctx->device);
    LaunchTimer lt(ctx, (hipStream_t)stream);
    HIPCHK(hipMemcpyAsync(dev_state, &ra.rk, sizeof(RngKey), hipMemcpyHostToDevice, (hipStream_t)stream));
    HIPCHK(hipStreamSynchronize((hipStream_t)stream));     // the source is on this stack frame
    return FFGPU_OK;
}

int ffgpu_split_rng_state(ffgpu_ctx* ctx, const void* secrets, const void* mul_by, void* dev_state, int t, int m,
                          void* shares, size_t share_stride, size_t n, void* stream) {
    ARGCHK(ctx && dev_state);
    ARGCHK(m >= 1 && t >= 0 && t < m);
    if (n == 0) return FFGPU_OK;
    ARGCHK(secrets && shares && (m == 1 || share_stride >= n));
    RngArgs ra;
    memset(&ra, 0, sizeof(ra));
    ra.rk.rounds = 20;
    ra.r0 = ctx->rng_r[0];
    ra.r1 = ctx->rng_r[1];
    ra.dev_key = (const RngKey*)dev_state;
    DeviceGuard g(ctx->device);
    LaunchTimer lt(ctx, (hipStream_t)stream);
    if (mul_by && t > 0) ra.aux = gf8_tables_on_device(ctx);
    int rc = ctx->ops->split(ctx->policy, ctx->device, secrets, mul_by, nullptr, 0, t, m, shares, share_stride, n,
                             (hipStream_t)stream, t > 0 ? &ra : nullptr);
    if (rc) return launch_status(rc);
    return FFGPU_OK;       // the kernel's last workgroup advanced the nonce (rng_state_release)
}

int ffgpu_gate_rng(ffgpu_ctx* ctx, const void* const* host_rows_a, const uint64_t* host_lambda_a, int ka,
                   const void* const* host_rows_b, const uint64_t* host_lambda_b, int kb, const uint8_t* host_key32,
                   uint64_t nonce, int rounds, void* dev_state, int t, int m, void* shares, size_t share_stride,
                   size_t n, void* stream) {
    return ffgpu_gate_rng_batch(ctx, host_rows_a, host_lambda_a, ka, 0, host_rows_b, host_lambda_b, kb, 0, host_key32,
                                dev_state ? 0 : nonce, rounds, dev_state, 0, t, m, shares, share_stride, 0, n, 1, stream);
}

int ffgpu_rng_state_advance(ffgpu_ctx* ctx, void* dev_state, uint32_t by, void* stream) {
    ARGCHK(ctx && dev_state);
    if (by == 0) return FFGPU_OK;
    DeviceGuard g(ctx->device);
    hipLaunchKernelGGL((k_rng_advance<0>), dim3(1), dim3(1), 0, (hipStream_t)stream, (RngKey*)dev_state, by);
    return hipGetLastError() == hipSuccess ? FFGPU_OK : FFGPU_EHIP;
}

int ffgpu_gate_rng_batch(ffgpu_ctx* ctx, const void* const* host_rows_a, const uint64_t* host_lambda_a, int ka,
                         size_t batch_stride_a, const void* const* host_rows_b, const uint64_t* host_lambda_b, int kb,
                         size_t batch_stride_b, const uint8_t* host_key32, uint64_t nonce, int rounds, void* dev_state,
                         int defer_advance, int t, int m, void* shares, size_t share_stride, size_t batch_stride_out,
                         size_t n, int nbatch, void* stream) {
    ARGCHK(ctx);
    ARGCHK(nbatch >= 1 && nbatch <= 255);
    ARGCHK(!dev_state || nonce <= 0xffffffffull);
    // host-key path: the batch row goes into bits 40..47 of the 64-bit nonce (bits 8..15 of nonce word 1); a caller
    // nonce that reaches those bits would alias another row's generator stream under a reused key
    ARGCHK(dev_state || nbatch == 1 || nonce < (1ull << 40));
    ARGCHK(m >= 1 && t >= 1 && t < m && ka >= 1 && kb >= 0);
    if (t > 3 || ka > 7 || kb > 7) return FFGPU_ENOTSUP;
    ARGCHK(host_rows_a && host_lambda_a && (kb == 0 || (host_rows_b && host_lambda_b)));
    RngArgs ra;
    if (dev_state) {
        memset(&ra, 0, sizeof(ra));
        ra.rk.rounds = 20;
        ra.r0 = ctx->rng_r[0];
        ra.r1 = ctx->rng_r[1];
        ra.dev_key = (const RngKey*)dev_state;
        ra.nonce_off = (uint32_t)nonce;
        ra.no_advance = defer_advance ? 1 : 0;
    } else {
        int rc = make_rng(ctx, host_key32, nonce, rounds, &ra);
        if (rc != FFGPU_OK) return rc;
    }
    if (n == 0) return FFGPU_OK;
    ARGCHK(shares && (m == 1 || share_stride >= n));
    for (int j = 0; j < ka; ++j) ARGCHK(host_rows_a[j]);
    for (int j = 0; j < kb; ++j) ARGCHK(host_rows_b[j]);
    DeviceGuard g(ctx->device);
    LaunchTimer lt(ctx, (hipStream_t)stream);
    ra.aux = gf8_tables_on_device(ctx);
    return launch_status(ctx->ops->gate(ctx->policy, ctx->device, host_rows_a, host_lambda_a, ka, host_rows_b,
                                        host_lambda_b, kb, t, m, shares, share_stride, n, (hipStream_t)stream, &ra,
                                        nbatch, batch_stride_a, batch_stride_b, batch_stride_out));
}

int ffgpu_split(ffgpu_ctx* ctx, const void* secrets, const void* coeffs, size_t coeff_stride, int t, int m,
                void* shares, size_t share_stride, size_t n, void* stream) {
    return do_split(ctx, secrets, nullptr, false, coeffs, coeff_stride, t, m, shares, share_stride, n, stream);
}
int ffgpu_mul_split(ffgpu_ctx* ctx, const void* a, const void* b, const void* coeffs, size_t coeff_stride,
                    int t, int m, void* shares, size_t share_stride, size_t n, void* stream) {
    return do_split(ctx, a, b, true, coeffs, coeff_stride, t, m, shares, share_stride, n, stream);
}

int ffgpu_rng_coeffs(ffgpu_ctx* ctx, const uint8_t* host_key32, uint64_t nonce, int rounds, int t, void* coeffs,
                     size_t coeff_stride, size_t n, void* stream) {
    ARGCHK(ctx);
    ARGCHK(t >= 1);
    RngArgs ra;
    int rc = make_rng(ctx, host_key32, nonce, rounds, &ra);
    if (rc != FFGPU_OK) return rc;
    if (n == 0) return FFGPU_OK;
    ARGCHK(coeffs && (t == 1 || coeff_stride >= n));
    DeviceGuard g(ctx->device);
    LaunchTimer lt(ctx, (hipStream_t)stream);
    return launch_status(ctx->ops->rng_coeffs(ctx->policy, ctx->device, coeffs, coeff_stride, t, n,
                                              (hipStream_t)stream, &ra));
}
int ffgpu_split_rng(ffgpu_ctx* ctx, const void* secrets, const uint8_t* host_key32, uint64_t nonce, int rounds,
                    int t, int m, void* shares, size_t share_stride, size_t n, void* stream) {
    return do_split_rng(ctx, secrets, nullptr, false, host_key32, nonce, rounds, t, m, shares, share_stride, n,
                        stream);
}
int ffgpu_mul_split_rng(ffgpu_ctx* ctx, const void* a, const void* b, const uint8_t* host_key32, uint64_t nonce,
                        int rounds, int t, int m, void* shares, size_t share_stride, size_t n, void* stream) {
    return do_split_rng(ctx, a, b, true, host_key32, nonce, rounds, t, m, shares, share_stride, n, stream);
}

int ffgpu_recombine(ffgpu_ctx* ctx, const void* const* host_rows, const uint64_t* host_lambda, int k, int w,
                    void* out, size_t out_stride, size_t n, void* stream) {
    ARGCHK(ctx);
    ARGCHK(k >= 1 && w >= 1);
    if (n == 0) return FFGPU_OK;
    ARGCHK(host_rows && host_lambda && out && (w == 1 || out_stride >= n));
    for (int j = 0; j < k; ++j) ARGCHK(host_rows[j]);
    DeviceGuard g(ctx->device);
    LaunchTimer lt(ctx, (hipStream_t)stream);
    // GF(2^n), 9 <= n <= 128 with a sparse modulus: shared nibble tables of the (uniform) Lagrange
    // coefficients in LDS instead of one full field multiplication per row and element
    if ((ctx->policy_kind == POL_GF2W64 || ctx->policy_kind == POL_GF2W128) && !ctx->gf2w_limbs && k <= 9 &&
        n >= 65536) {
        const int limbs = ctx->policy_kind == POL_GF2W128 ? 2 : 1;
        int rc = 0;
        for (int r = 0; r < w && rc == 0; ++r)
            rc = ffgpu_launch_gf2w_recombine(ctx->policy, limbs, ctx->device, host_rows, host_lambda + 2 * (size_t)r * k,
                                             k, (char*)out + (size_t)r * out_stride * ctx->elem_bytes, n,
                                             (hipStream_t)stream);
        if (rc != 2) return launch_status(rc);
    }
    return launch_status(ctx->ops->recombine(ctx->policy, ctx->device, host_rows, host_lambda, k, w, out,
                                             out_stride, n, (hipStream_t)stream));
}

int ffgpu_matmul(ffgpu_ctx* ctx, const void* A, size_t lda, const void* B, size_t ldb, void* C, size_t ldc,
                 size_t M, size_t K, size_t N, void* stream) {
    ARGCHK(ctx);
    if (M == 0 || N == 0) return FFGPU_OK;
    ARGCHK(C && ldc >= N && M < (1u << 30) && N < (1u << 30) && K < (1u << 30));
    ARGCHK(K == 0 || (A && B && lda >= K && ldb >= N));
    DeviceGuard g(ctx->device);
    LaunchTimer lt(ctx, (hipStream_t)stream);
    void* ws = nullptr;
    size_t ws_bytes = 0;
    {
        // scratch: int8 limb planes for the matrix-core product (10 x (M + N) x K bytes, up to 8 GiB), else 64 MiB of
        // split-K partial sums
        size_t want = 0;
        const size_t Mp = (M + 63) / 64 * 64, Np = (N + 63) / 64 * 64, Kp = (K + 31) / 32 * 32;
        if (ctx->kind == FFGPU_PRIME && M > 8 && N > 8 && K >= 64 && ctx->tune.mm_mfma && (double)M * N * K >= ctx->tune.mm_mfma_min) {
            want = (size_t)(ctx->elem_bytes <= 8 ? 8 : 16) * (Mp + Np) * Kp + ((size_t)64 << 20);   // digit planes per operand + split-K slabs
            if (want > ((size_t)8 << 30)) want = 0;
        }
        if (!want && K >= 64 && ((M + 31) / 32) * ((N + 31) / 32) < 2048) want = (size_t)64 << 20;
        if (want) {
            std::lock_guard<std::mutex> lk(*ctx->scratch_mu);
            ffgpu_ctx::Scratch* slot = nullptr;
            for (auto& sc : ctx->scratch)
                if (sc.used && sc.stream == (hipStream_t)stream) slot = &sc;
            if (!slot) {
                for (auto& sc : ctx->scratch)
                    if (!sc.used && !slot) slot = &sc;
                if (!slot) {                                     // more streams than slots: recycle the smallest buffer
                    slot = &ctx->scratch[0];
                    for (auto& sc : ctx->scratch)
                        if (sc.bytes < slot->bytes) slot = &sc;
                    HIPCHK(hipStreamSynchronize(slot->stream));
                    if (slot->ptr) (void)hipFree(slot->ptr);
                    slot->ptr = nullptr;
                    slot->bytes = 0;
                }
                slot->used = 1;
                slot->stream = (hipStream_t)stream;
            }
            if (slot->bytes < want) {
                HIPCHK(hipStreamSynchronize((hipStream_t)stream));   // queued work may still read the old buffer
                if (slot->ptr) (void)hipFree(slot->ptr);
                slot->ptr = nullptr;
                slot->bytes = 0;
                if (hipMalloc(&slot->ptr, want) == hipSuccess) slot->bytes = want;
                else (void)hipGetLastError();                     // no scratch: the VALU kernels need none
            }
            ws = slot->ptr;
            ws_bytes = slot->bytes;
        }
    }
    return launch_status(ctx->ops->matmul(ctx->policy, ctx->device, A, lda, B, ldb, C, ldc, (int)M, (int)K, (int)N, ws,
                                          ws_bytes, &ctx->tune, (hipStream_t)stream));
}

int ffgpu_group_matvec(ffgpu_ctx* ctx, const uint64_t* host_matrix, const uint64_t* host_bias, int r, int g,
                       const void* in, void* out, size_t ngroups, void* stream) {
    ARGCHK(ctx && host_matrix && r >= 1 && g >= 1);
    if (r > 16 || g > 16) return FFGPU_ENOTSUP;
    if (ngroups == 0) return FFGPU_OK;
    ARGCHK(in && out);
    DeviceGuard gd(ctx->device);
    LaunchTimer lt(ctx, (hipStream_t)stream);
    if (ctx->kind == FFGPU_BINARY && ctx->elem_bytes == 1 && g == 8 && (((uintptr_t)in) & 7u) == 0) {
        // groups of 8 bytes: packed-byte kernel (misc.hip)
        if (r == 8 && (((uintptr_t)out) & 7u) == 0)
            return launch_status(ffgpu_launch_gf8_group8(ctx->policy, ctx->device, host_matrix, host_bias, 0, in, out,
                                                         ngroups, (hipStream_t)stream));
        bool pow2 = (r == 1);
        for (int c = 0; pow2 && c < 8; ++c) pow2 = (host_matrix[2 * c] == (1ull << c));
        if (pow2 && (!host_bias || (host_bias[0] & 0xffu) == 0)) {   // np_from_bits: identity, then fold by 2^r
            uint64_t eye[128];
            memset(eye, 0, sizeof(eye));
            for (int c = 0; c < 8; ++c) eye[2 * (c * 8 + c)] = 1;
            return launch_status(ffgpu_launch_gf8_group8(ctx->policy, ctx->device, eye, nullptr, 1, in, out, ngroups,
                                                         (hipStream_t)stream));
        }
    }
    return launch_status(ctx->ops->group_matvec(ctx->policy, ctx->device, host_matrix, host_bias, r, g, in, out,
                                                ngroups, (hipStream_t)stream));
}

int ffgpu_gf256_bit_affine(ffgpu_ctx* ctx, const uint64_t* host_matrix, const uint64_t* host_bias, int from_bits,
                           const void* in, void* out, size_t n, void* stream) {
    ARGCHK(ctx && host_matrix);
    if (ctx->kind != FFGPU_BINARY || ctx->elem_bytes != 1) return FFGPU_ENOTSUP;
    if (n == 0) return FFGPU_OK;
    ARGCHK(in && out);
    if ((((uintptr_t)in) & 7u) || (!from_bits && (((uintptr_t)out) & 7u))) return FFGPU_EINVAL;
    DeviceGuard gd(ctx->device);
    LaunchTimer lt(ctx, (hipStream_t)stream);
    return launch_status(ffgpu_launch_gf8_group8(ctx->policy, ctx->device, host_matrix, host_bias, from_bits ? 1 : 0, in,
                                                 out, n, (hipStream_t)stream));
}

static int do_dot(ffgpu_ctx* ctx, const void* a, const void* b, void* out, void* workspace, size_t n, void* stream) {
    ARGCHK(ctx && out);
    DeviceGuard g(ctx->device);
    LaunchTimer lt(ctx, (hipStream_t)stream);
    if (n == 0) {   // empty sum = 0
        HIPCHK(hipMemsetAsync(out, 0, (size_t)ctx->elem_bytes, (hipStream_t)stream));
        return FFGPU_OK;
    }
    ARGCHK(a && workspace);
    return launch_status(ctx->ops->dot(ctx->policy, ctx->device, a, b, out, workspace, n, (hipStream_t)stream));
}
int ffgpu_dot(ffgpu_ctx* ctx, const void* a, const void* b, void* out, void* workspace, size_t n, void* stream) {
    ARGCHK(n == 0 || b);
    return do_dot(ctx, a, b, out, workspace, n, stream);
}
int ffgpu_sum(ffgpu_ctx* ctx, const void* a, void* out, void* workspace, size_t n, void* stream) {
    return do_dot(ctx, a, nullptr, out, workspace, n, stream);
}

int ffgpu_prss_combine(ffgpu_ctx* ctx, const void* const* host_streams, int ks, int d, int l, int mask_bits,
                       const uint64_t* host_weights, int accumulate, void* out, size_t n, void* stream) {
    ARGCHK(ctx);
    ARGCHK(ks >= 1 && d >= 1 && l >= 1 && l <= 64 && mask_bits >= 0 && mask_bits <= 192);
    if (n == 0) return FFGPU_OK;
    ARGCHK(host_streams && host_weights && out);
    for (int s = 0; s < ks; ++s) ARGCHK(host_streams[s]);
    // limb radix constant for the wide reduction: 2^(8*elem_bytes) mod p (prime policies)
    uint64_t r2[2] = {ctx->rng_r[0], ctx->rng_r[1]};
    DeviceGuard g(ctx->device);
    LaunchTimer lt(ctx, (hipStream_t)stream);
    return launch_status(ctx->ops->prss(ctx->policy, ctx->device, host_streams, ks, d, l, mask_bits, host_weights,
                                        r2, accumulate, out, n, (hipStream_t)stream));
}

int ffgpu_prss_chacha(ffgpu_ctx* ctx, const uint8_t* host_keys, int ks, int d, int l, int mask_bits, int rounds,
                      const uint64_t* host_weights, int accumulate, void* out, size_t n, void* stream) {
    ARGCHK(ctx);
    ARGCHK(ks >= 1 && d >= 1 && l >= 1 && l <= 64 && mask_bits >= 0 && mask_bits <= 192);
    ARGCHK(rounds == 20 || rounds == 12 || rounds == 8);
    if (n == 0) return FFGPU_OK;
    ARGCHK(host_keys && host_weights && out);
    uint64_t r2[2] = {ctx->rng_r[0], ctx->rng_r[1]};
    DeviceGuard g(ctx->device);
    LaunchTimer lt(ctx, (hipStream_t)stream);
    return launch_status(ctx->ops->prss_chacha(ctx->policy, ctx->device, host_keys, ks, d, l, mask_bits, rounds, host_weights,
                                               r2, accumulate, out, n, (hipStream_t)stream));
}
int ffgpu_prss_chacha_layout(int l, int* tb, int* dpt) {
    ARGCHK(l >= 1 && l <= 64 && tb && dpt);
    ffgpu::prss_cc_layout(l, tb, dpt);
    return FFGPU_OK;
}

int ffgpu_gf256_to_bits(ffgpu_ctx* ctx, const void* in, const void* addend, void* out, size_t n, void* stream) {
    ARGCHK(ctx);
    if (ctx->kind != FFGPU_BINARY || ctx->elem_bytes != 1) return FFGPU_ENOTSUP;
    if (n == 0) return FFGPU_OK;
    ARGCHK(in && out);
    DeviceGuard g(ctx->device);
    LaunchTimer lt(ctx, (hipStream_t)stream);
    return launch_status(ffgpu_launch_gf8_to_bits(ctx->device, in, addend, out, n, (hipStream_t)stream));
}

int ffgpu_gf256_mask_open(ffgpu_ctx* ctx, const void* const* host_rows, const uint64_t* host_coef, int nrows,
                          const void* const* host_rbits, const uint64_t* host_mu, int np, void* out, size_t n, void* stream) {
    ARGCHK(ctx);
    if (ctx->kind != FFGPU_BINARY || ctx->elem_bytes != 1) return FFGPU_ENOTSUP;
    ARGCHK(nrows >= 0 && np >= 0 && nrows + np >= 1);
    if (nrows > 32 || np > 8) return FFGPU_ENOTSUP;
    if (n == 0) return FFGPU_OK;
    ARGCHK(out && (nrows == 0 || (host_rows && host_coef)) && (np == 0 || (host_rbits && host_mu)));
    for (int r = 0; r < nrows; ++r) ARGCHK(host_rows[r]);
    for (int p = 0; p < np; ++p) ARGCHK(host_rbits[p]);
    DeviceGuard g(ctx->device);
    LaunchTimer lt(ctx, (hipStream_t)stream);
    return launch_status(ffgpu_launch_gf8_mask_open(ctx->policy, ctx->device, host_rows, host_coef, nrows, host_rbits, host_mu,
                                                    np, out, n, (hipStream_t)stream));
}

int ffgpu_gf256_bits_affine_fold(ffgpu_ctx* ctx, const uint64_t* host_matrix, const uint64_t* host_bias, const void* c,
                                 const void* rbits, size_t rbits_batch_stride, void* out, size_t out_batch_stride, size_t n,
                                 int nbatch, void* stream) {
    ARGCHK(ctx && host_matrix);
    if (ctx->kind != FFGPU_BINARY || ctx->elem_bytes != 1) return FFGPU_ENOTSUP;
    ARGCHK(nbatch >= 1 && nbatch <= 65535);
    if (n == 0) return FFGPU_OK;
    ARGCHK(c && rbits && out);
    ARGCHK(nbatch == 1 || (rbits_batch_stride >= 8 * n && out_batch_stride >= n));
    DeviceGuard g(ctx->device);
    LaunchTimer lt(ctx, (hipStream_t)stream);
    return launch_status(ffgpu_launch_gf8_bits_affine_fold(ctx->policy, ctx->device, host_matrix, host_bias, c, rbits,
                                                           rbits_batch_stride, out, out_batch_stride, n, nbatch,
                                                           (hipStream_t)stream));
}

int ffgpu_gf256_sbox_layer(ffgpu_ctx* ctx, const uint64_t* host_matrix, const uint64_t* host_bias, const uint64_t* host_lambda,
                           const uint64_t* host_mu, int t, int m, const void* x, size_t x_stride, const void* rbits,
                           size_t rbits_stride, void* out, size_t out_stride, size_t n, const uint8_t* host_key32, uint64_t nonce,
                           int rounds, void* dev_state, int defer_advance, void* stream) {
    ARGCHK(ctx && host_matrix && host_lambda && host_mu);
    if (ctx->kind != FFGPU_BINARY || ctx->elem_bytes != 1 || !ctx->gf8_tab_min) return FFGPU_ENOTSUP;
    ARGCHK(t >= 1 && m >= 2 * t + 1);
    if (t > 3 || m > 7) return FFGPU_ENOTSUP;
    ARGCHK(!dev_state || nonce <= 0xffffffffull);
    RngArgs ra;
    if (dev_state) {
        memset(&ra, 0, sizeof(ra));
        ra.rk.rounds = 20;
        ra.dev_key = (const RngKey*)dev_state;
        ra.nonce_off = (uint32_t)nonce;
        ra.no_advance = defer_advance ? 1 : 0;
    } else {
        int rc = make_rng(ctx, host_key32, nonce, rounds, &ra);
        if (rc != FFGPU_OK) return rc;
    }
    if (n == 0) return FFGPU_OK;
    ARGCHK(x && rbits && out && x_stride >= n && out_stride >= n && rbits_stride >= 8 * n);
    DeviceGuard g(ctx->device);
    // tables: log / antilog of the field, np_from_bits, affine fold -- cached on the device per (matrix, bias)
    unsigned char key[72];
    for (int i = 0; i < 64; ++i) key[i] = (unsigned char)(host_matrix[2 * i] & 0xffu);
    for (int i = 0; i < 8; ++i) key[64 + i] = host_bias ? (unsigned char)(host_bias[2 * i] & 0xffu) : 0;
    {
        static std::mutex mu_;
        std::lock_guard<std::mutex> lk(mu_);
        if (!ctx->sbl_tables_dev) {
            void* d = nullptr;
            if (hipMalloc(&d, 1536 + 2304 + 2304) != hipSuccess) return FFGPU_ENOMEM;
            ctx->sbl_tables_dev = d;
            ctx->sbl_valid = 0;
        }
        if (!ctx->sbl_valid || memcmp(key, ctx->sbl_key, sizeof(key)) != 0) {
            unsigned char host_tables[1536 + 2304 + 2304];
            ffgpu_gf8_sbox_layer_tables(ctx->policy, ctx->gf8_tables, host_matrix, host_bias, host_tables);
            // a DIFFERENT affine map than the cached one: launches that still read the old tables (on any stream) finish first
            if (ctx->sbl_valid) HIPCHK(hipDeviceSynchronize());
            // (synchronous copy: not inside a stream capture -- engine.CapturedLaunches warms the call up first)
            HIPCHK(hipMemcpy(ctx->sbl_tables_dev, host_tables, sizeof(host_tables), hipMemcpyHostToDevice));
            memcpy(ctx->sbl_key, key, sizeof(key));
            ctx->sbl_valid = 1;
        }
    }
    LaunchTimer lt(ctx, (hipStream_t)stream);
    return launch_status(ffgpu_launch_gf8_sbox_layer(ctx->policy, ctx->device, x, x_stride, rbits, rbits_stride, out, out_stride,
                                                     ctx->sbl_tables_dev, host_lambda, host_mu, t, m, n, (hipStream_t)stream, &ra));
}

int ffgpu_gf256_sbox(ffgpu_ctx* ctx, const void* in, const uint8_t* host_rows8, uint8_t b, void* out,
                     size_t n, void* stream) {
    ARGCHK(ctx && host_rows8);
    if (ctx->kind != FFGPU_BINARY || ctx->elem_bytes != 1) return FFGPU_ENOTSUP;
    GF2P8 f;
    memcpy(&f, ctx->policy, sizeof(f));
    if (f.n != 8) return FFGPU_ENOTSUP;
    if (n == 0) return FFGPU_OK;
    ARGCHK(in && out);
    uint8_t lut[256];
    {
        static std::mutex mu;
        std::lock_guard<std::mutex> lk(mu);
        uint8_t key[9];
        memcpy(key, host_rows8, 8);
        key[8] = b;
        if (!ctx->sbox_valid || memcmp(key, ctx->sbox_key, 9) != 0) {
            ffgpu_sbox_build_lut(ctx->policy, host_rows8, b, ctx->sbox_lut);
            memcpy(ctx->sbox_key, key, 9);
            ctx->sbox_valid = 1;
        }
        memcpy(lut, ctx->sbox_lut, 256);
    }
    DeviceGuard g(ctx->device);
    LaunchTimer lt(ctx, (hipStream_t)stream);
    return launch_status(ffgpu_launch_sbox(lut, ctx->device, in, out, n, (hipStream_t)stream));
}

}  // extern "C"

// ---- timing helpers: HIP events on the launch stream ------------------------
template <class Fn>
static int time_loop(ffgpu_ctx* ctx, int reps, void* stream, float* ms, Fn fn) {
    ARGCHK(ctx && ms && reps >= 1);
    DeviceGuard g(ctx->device);
    hipStream_t st = (hipStream_t)stream;
    hipEvent_t e0, e1;
    HIPCHK(hipEventCreate(&e0));
    HIPCHK(hipEventCreate(&e1));
    int rc = FFGPU_OK;
    hipError_t he = hipEventRecord(e0, st);
    for (int i = 0; i < reps && rc == FFGPU_OK && he == hipSuccess; ++i) rc = fn();
    if (he == hipSuccess) he = hipEventRecord(e1, st);
    if (he == hipSuccess) he = hipEventSynchronize(e1);
    float t = 0.f;
    if (he == hipSuccess) he = hipEventElapsedTime(&t, e0, e1);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    if (rc != FFGPU_OK) return rc;
    if (he != hipSuccess) return hip_fail(he, "event timing");
    *ms = t / (float)reps;
    return FFGPU_OK;
}

extern "C" {

int ffgpu_time_mul(ffgpu_ctx* ctx, const void* a, const void* b, void* out, size_t n, int reps, void* stream,
                   float* ms) {
    return time_loop(ctx, reps, stream, ms, [&]() { return ffgpu_mul(ctx, a, b, out, n, stream); });
}
int ffgpu_time_split(ffgpu_ctx* ctx, const void* secrets, const void* coeffs, size_t coeff_stride, int t,
                     int m, void* shares, size_t share_stride, size_t n, int reps, void* stream, float* ms) {
    return time_loop(ctx, reps, stream, ms, [&]() {
        return ffgpu_split(ctx, secrets, coeffs, coeff_stride, t, m, shares, share_stride, n, stream);
    });
}
int ffgpu_time_recombine(ffgpu_ctx* ctx, const void* const* host_rows, const uint64_t* host_lambda, int k,
                         int w, void* out, size_t out_stride, size_t n, int reps, void* stream, float* ms) {
    return time_loop(ctx, reps, stream, ms, [&]() {
        return ffgpu_recombine(ctx, host_rows, host_lambda, k, w, out, out_stride, n, stream);
    });
}
int ffgpu_copy(ffgpu_ctx* ctx, const void* src, void* dst, size_t bytes, void* stream) {
    ARGCHK(ctx);
    if (bytes == 0) return FFGPU_OK;
    ARGCHK(src && dst);
    DeviceGuard g(ctx->device);
    LaunchTimer lt(ctx, (hipStream_t)stream);
    return launch_status(ffgpu_launch_copy(ctx->device, src, dst, bytes, (hipStream_t)stream));
}
int ffgpu_valu_probe(ffgpu_ctx* ctx, int op, int iters, int waves_per_simd, void* scratch32, double* out3, void* stream) {
    ARGCHK(ctx && scratch32 && out3);
    ARGCHK(op >= 0 && op <= 13 && iters >= 1 && waves_per_simd >= 1 && waves_per_simd <= 8);
    DeviceGuard g(ctx->device);
    return launch_status(ffgpu_launch_valu_probe(ctx->device, op, iters, waves_per_simd, scratch32, out3, (hipStream_t)stream));
}
int ffgpu_time_copy(ffgpu_ctx* ctx, const void* src, void* dst, size_t bytes, int reps, void* stream,
                    float* ms) {
    ARGCHK(ctx && src && dst);
    return time_loop(ctx, reps, stream, ms, [&]() {
        return launch_status(ffgpu_launch_copy(ctx->device, src, dst, bytes, (hipStream_t)stream));
    });
}

}  // extern "C"
