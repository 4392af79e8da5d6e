// launch.hpp -- host side of the kernels: per-policy launcher table (FieldOps) behind the C ABI of api.hip.
// Included by kernels.hpp.
#pragma once

namespace ffgpu {

// ---- launch plumbing -------------------------------------------------------
struct LaunchCfg {
    int blocks_per_cu;  // 0 = uncapped grid: one 16-byte pack per thread (default, measured best)
    int num_cu;
};
LaunchCfg launch_cfg(int device);

// The library's run-time switches (one table: INTEGRATION.md section 6).  Read from the environment ONCE per context, when
// it is created (ffgpu_ctx_create), and carried by it: no getenv on any call path.
struct Tuning {
    int mm_mfma;            // FFGPU_MM_MFMA=0: dense products stay on the VALU kernel (cross-checks of the matrix-core path)
    double mm_mfma_min;     // FFGPU_MM_MFMA_MIN: smallest M*N*K that goes to the matrix cores (default 8e7)
    int gf2w_bitsliced;     // FFGPU_GF2W_BITSLICED=0: GF(2^64) products through the multiplier kernel only
};

inline unsigned grid_for(size_t iters, const LaunchCfg& lc) {
    size_t want = (iters + BLOCK - 1) / BLOCK;
    size_t cap = lc.blocks_per_cu > 0 ? (size_t)lc.blocks_per_cu * (size_t)lc.num_cu : (size_t)0x7fffffff;
    if (want < 1) want = 1;
    return (unsigned)(want < cap ? want : cap);
}

inline bool aligned16(const void* p) { return ((uintptr_t)p & 15u) == 0; }

// host-side table of launchers for one policy type; the context stores the
// policy blob and a pointer to this table.
struct FieldOps {
    int (*ew2)(const void* F, int device, int op, const void* a, const void* b, void* o, size_t n,
               hipStream_t st);
    int (*ew1)(const void* F, int device, int op, const void* a, const uint64_t* scalar2, void* o,
               size_t n, hipStream_t st);
    int (*muladd)(const void* F, int device, const void* a, const void* b, const void* c, void* o,
                  size_t n, hipStream_t st);
    // coef == nullptr && rng != nullptr: coefficients are drawn in-kernel from the keystream
    int (*split)(const void* F, int device, const void* a, const void* b, const void* coef,
                 size_t cstride, int t, int m, void* out, size_t ostride, size_t n, hipStream_t st,
                 const RngArgs* rng);
    int (*rng_coeffs)(const void* F, int device, void* coef, size_t cstride, int t, size_t n, hipStream_t st,
                      const RngArgs* rng);
    int (*recombine)(const void* F, int device, const void* const* rows, const uint64_t* lam2, int k,
                     int w, void* out, size_t ostride, size_t n, hipStream_t st);
    int (*pow)(const void* F, int device, const void* a, const ExpArgs* ex, void* out, size_t n, hipStream_t st);
    int (*inv)(const void* F, int device, const void* a, const ExpArgs* ex, void* out, size_t n, int* flag,
               hipStream_t st);
    int (*matmul)(const void* F, int device, const void* A, size_t lda, const void* B, size_t ldb, void* C,
                  size_t ldc, int M, int K, int N, void* workspace, size_t workspace_bytes, const Tuning* tune,
                  hipStream_t st);
    int (*dot)(const void* F, int device, const void* a, const void* b, void* out, void* workspace, size_t n,
               hipStream_t st);
    // nbatch > 1: gridDim.y independent gates in one launch, operands / outputs of gate y at element offsets y*yA, y*yB, y*yO
    int (*gate)(const void* F, int device, const void* const* rowsA, const uint64_t* lamA2, int kA,
                const void* const* rowsB, const uint64_t* lamB2, int kB, int t, int m, void* out, size_t ostride,
                size_t n, hipStream_t st, const RngArgs* rng, int nbatch, size_t yA, size_t yB, size_t yO);
    int (*sqrt_cl)(const void* F, int device, const void* a, const ExpArgs* eleg, const ExpArgs* elad, void* out, size_t n,
                   hipStream_t st);
    int (*gauss)(const void* F, int device, void* A, int n, int ncols, size_t batch, int det_mode, const ExpArgs* ex,
                 void* det, int* sing, hipStream_t st);
    int (*group_matvec)(const void* F, int device, const uint64_t* m2, const uint64_t* bias2, int r, int g,
                        const void* in, void* out, size_t ngroups, hipStream_t st);
    int (*beaver)(const void* F, int device, const void* z, const void* x, const void* y, const void* d, const void* e,
                  void* out, int add_de, size_t n, hipStream_t st);
    int (*prss)(const void* F, int device, const void* const* streams, int ks, int d, int l, int mask_bits,
                const uint64_t* weights2, const uint64_t* r2, int accumulate, void* out, size_t n, hipStream_t st);
    // keys40: ks x (32-byte ChaCha key + 8-byte nonce)
    int (*prss_chacha)(const void* F, int device, const uint8_t* keys40, int ks, int d, int l, int mask_bits, int rounds,
                       const uint64_t* weights2, const uint64_t* r2, int accumulate, void* out, size_t n, hipStream_t st);
};

// Host scalars (Lagrange coefficients, constants, matrix entries) cross the C ABI as little-endian 64-bit limbs:
// 2 per scalar, 3 for the three-limb prime fields (ffgpu_ctx_scalar_limbs).
template <class F>
constexpr int scalar_limbs() {
    return sizeof(typename F::word) == 24 ? 3 : 2;
}
// scalar number idx of a host array -> policy word (broadcast for packed fields)
template <class F>
inline typename F::word word_at(const F& f, const uint64_t* base, size_t idx);
template <class F>
inline typename F::word word_from_limbs(const F& f, uint64_t lo, uint64_t hi) {
    if constexpr (sizeof(typename F::word) == 24) {
        typename F::word w;
        w.lo = lo;
        w.mid = hi;
        w.hi = 0;
        return w;
    } else if constexpr (sizeof(typename F::word) == 16) {
        typename F::word w;
        w.lo = lo;
        w.hi = hi;
        return w;
    } else if constexpr (F::EPW == 4) {
        uint32_t b = (uint32_t)(lo & 0xffu);
        return b * 0x01010101u;
    } else {
        return (typename F::word)lo;
    }
}
template <class F>
inline typename F::word word_at(const F& f, const uint64_t* base, size_t idx) {
    constexpr int SL = scalar_limbs<F>();
    const uint64_t* l = base + idx * SL;
    if constexpr (SL == 3) {
        typename F::word w;
        w.lo = l[0];
        w.mid = l[1];
        w.hi = l[2];
        return w;
    } else {
        return word_from_limbs<F>(f, l[0], l[1]);
    }
}

template <class F>
inline typename F::word prep_const(const F& f, typename F::word c) {
    return f.prep(c);
}

#define FFGPU_CHECK_LAUNCH()                      \
    do {                                          \
        hipError_t e__ = hipGetLastError();       \
        if (e__ != hipSuccess) return (int)e__ | 0x10000; \
    } while (0)

template <class F>
struct Launchers {
    typedef typename F::elem E;
    typedef typename F::word W;
    enum { EPV = Pack<W>::N * F::EPW };  // elements per pack (one lane's access)
    // dwordx3 needs dword alignment only; three-limb (24-byte) elements are moved by the WAVE as dwordx4 (kernels.hpp, ldgw)
    enum { PACK_ALIGN = sizeof(E) == 12 ? 4 : 16 };
    static bool al(const void* p) { return ((uintptr_t)p & (PACK_ALIGN - 1)) == 0; }
    // packs that go through the vector loop of a kernel (the rest: its scalar tail).  24-byte elements: whole waves only --
    // in `for (i = gid; i < nvec; i += gsz)` every wave is then entirely in or entirely out, which ldgw / stgw rely on
    static size_t nvec_of(size_t n, bool vec) {
        if (!vec) return 0;
        if (sizeof(E) == 24) return n & ~(size_t)63;
        return n / EPV;
    }
    // (a member function, not a lambda inside `matmul`: clang does not emit the host stub of a kernel specialisation
    // that is only named inside a generic lambda's discarded-branch neighbourhood)
    template <bool BRAW>
    static void launch_glds(const F& f, dim3 grid, hipStream_t st, const int8_t* Ap, const int8_t* Bp, E* out, size_t out_ld, int M, int N,
                            int Kp, int kb, int ke, int acc_, int kslice, size_t zs, const E* Braw, size_t ldb, uint64_t pmod) {
        if constexpr (F::EPW == 1 && !F::BINARY && sizeof(E) == 8) {
            const size_t lds = (size_t)(BRAW ? 8 : 6) * GLDS_TILE;
            static bool attr_done = false;
            if (!attr_done) {
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_limb_gemm_glds<F, BRAW>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                attr_done = true;
            }
            hipLaunchKernelGGL((k_limb_gemm_glds<F, BRAW>), grid, dim3(BLOCK), lds, st, f, Ap, Bp, out, out_ld, M, N, Kp, kb, ke, acc_,
                               kslice, zs, Braw, ldb, pmod);
        }
    }
    static bool stride_ok(size_t stride) { return (stride * sizeof(E)) % PACK_ALIGN == 0; }

    template <int OP>
    static void go_ew2(const F& f, const LaunchCfg& lc, const E* a, const E* b, E* o, size_t n, hipStream_t st) {
        bool vec = al(a) && al(b) && al(o);
        size_t nvec = nvec_of(n, vec);
        unsigned grid = grid_for(nvec ? nvec : n, lc);
        constexpr int OCC = EwOccupancy<F, OP>::waves;
        // (streamed loads and stores carry the non-temporal hint: +4-8 %, profiles/r01_tuning.md; the un-hinted instantiations
        // that rounds 1-5 kept behind FFGPU_NT=0 for A/B runs are gone)
        if constexpr (OCC > 0)
            hipLaunchKernelGGL((k_ew2_occ<F, OP, true, OCC>), dim3(grid), dim3(BLOCK), 0, st, f, a, b, o, nvec, n);
        else
            hipLaunchKernelGGL((k_ew2<F, OP, true>), dim3(grid), dim3(BLOCK), 0, st, f, a, b, o, nvec, n);
    }
    static int ew2(const void* Fp, int device, int op, const void* a, const void* b, void* o, size_t n,
                   hipStream_t st) {
        const F& f = *reinterpret_cast<const F*>(Fp);
        LaunchCfg lc = launch_cfg(device);
        const E* A = (const E*)a;
        const E* B = (const E*)b;
        E* O = (E*)o;
        switch (op) {
            case OP_ADD: go_ew2<OP_ADD>(f, lc, A, B, O, n, st); break;
            case OP_SUB: go_ew2<OP_SUB>(f, lc, A, B, O, n, st); break;
            case OP_MUL: go_ew2<OP_MUL>(f, lc, A, B, O, n, st); break;
            default: return 1;
        }
        FFGPU_CHECK_LAUNCH();
        return 0;
    }

    template <int OP>
    static void go_ew1(const F& f, const LaunchCfg& lc, const E* a, W s, E* o, size_t n, hipStream_t st) {
        bool vec = al(a) && al(o);
        size_t nvec = nvec_of(n, vec);
        unsigned grid = grid_for(nvec ? nvec : n, lc);
        hipLaunchKernelGGL((k_ew1<F, OP, true>), dim3(grid), dim3(BLOCK), 0, st, f, a, s, o, nvec, n);
    }
    static int ew1(const void* Fp, int device, int op, const void* a, const uint64_t* scalar2, void* o,
                   size_t n, hipStream_t st) {
        const F& f = *reinterpret_cast<const F*>(Fp);
        LaunchCfg lc = launch_cfg(device);
        W s = scalar2 ? word_at<F>(f, scalar2, 0) : word_from_limbs<F>(f, 0, 0);
        const E* A = (const E*)a;
        E* O = (E*)o;
        switch (op) {
            case OP_ADD: go_ew1<OP_ADD>(f, lc, A, s, O, n, st); break;
            case OP_RSUB: go_ew1<OP_RSUB>(f, lc, A, s, O, n, st); break;
            case OP_MUL: go_ew1<OP_MUL>(f, lc, A, s, O, n, st); break;
            case OP_NEG: go_ew1<OP_NEG>(f, lc, A, s, O, n, st); break;
            case OP_REDUCE: go_ew1<OP_REDUCE>(f, lc, A, s, O, n, st); break;
            case OP_COPY: go_ew1<OP_COPY>(f, lc, A, s, O, n, st); break;
            default: return 1;
        }
        FFGPU_CHECK_LAUNCH();
        return 0;
    }

    static int muladd(const void* Fp, int device, const void* a, const void* b, const void* c, void* o,
                      size_t n, hipStream_t st) {
        const F& f = *reinterpret_cast<const F*>(Fp);
        LaunchCfg lc = launch_cfg(device);
        bool vec = al(a) && al(b) && al(c) && al(o);
        size_t nvec = nvec_of(n, vec);
        unsigned grid = grid_for(nvec ? nvec : n, lc);
        hipLaunchKernelGGL((k_muladd<F, true>), dim3(grid), dim3(BLOCK), 0, st, f, (const E*)a,
                           (const E*)b, (const E*)c, (E*)o, nvec, n);
        FFGPU_CHECK_LAUNCH();
        return 0;
    }

    template <int T, bool FUSE, bool RNG, bool REC = false>
    static void go_split(const F& f, unsigned grid, bool nt, const E* a, const E* b, const E* coef,
                         size_t cstride, int m, E* out, size_t ostride, size_t nvec, size_t n, hipStream_t st,
                         const RngArgs& ra, const GateSrc<F>& gs, unsigned gy = 1) {
        (void)nt;
        hipLaunchKernelGGL((k_split<F, T, FUSE, true, RNG, REC>), dim3(grid, gy), dim3(BLOCK), 0, st, f, a, b,
                           coef, cstride, m, out, ostride, nvec, n, ra, gs);
    }
    template <bool FUSE, bool RNG>
    static int split_t(const F& f, const LaunchCfg& lc, const E* a, const E* b, const E* coef, size_t cstride,
                       int t, int m, E* out, size_t ostride, size_t n, hipStream_t st, const RngArgs& ra_in) {
        if (t > MAXT) {
            RngArgs ra = ra_in;
            unsigned grid = grid_for(n, lc);
            ra.release = !ra.no_advance && grid <= RNG_RELEASE_MAX_GRID;
            hipLaunchKernelGGL((k_split_any<F, FUSE, RNG>), dim3(grid), dim3(BLOCK), 0, st, f, a, b, coef, cstride,
                               t, m, out, ostride, n, ra);
            if (RNG && ra.dev_key && !ra.release && !ra.no_advance)
                hipLaunchKernelGGL((k_rng_advance<0>), dim3(1), dim3(1), 0, st, const_cast<RngKey*>(ra.dev_key), 1u);
            return 0;
        }
        bool vec = al(a) && (!FUSE || al(b)) && al(out) &&
                   (stride_ok(ostride) || m <= 1) &&
                   (RNG || t == 0 || (al(coef) && (stride_ok(cstride) || t <= 1)));
        size_t nvec = nvec_of(n, vec);
        // RNG kernels serve up to 4 packs per thread (RngLayout::G); a slightly larger grid is harmless.
        // Below ~2.6e5 packs the grouped loop cannot fill the chip: one pack per thread instead (ra.spread).
        RngArgs ra = ra_in;
        const bool spread = RNG && nvec > 0 && nvec < 262144;
        ra.spread = spread ? 1 : 0;
        unsigned grid = grid_for(nvec ? (RNG && !spread ? (n / EPV + 2) / 2 : nvec) : n, lc);
        ra.release = !ra.no_advance && grid <= RNG_RELEASE_MAX_GRID;
        const bool nt = true;
        GateSrc<F> gs;
        memset(&gs, 0, sizeof(gs));
        switch (t) {
            case 0: go_split<0, FUSE, false>(f, grid, nt, a, b, coef, cstride, m, out, ostride, nvec, n, st, ra, gs); break;
            case 1: go_split<1, FUSE, RNG>(f, grid, nt, a, b, coef, cstride, m, out, ostride, nvec, n, st, ra, gs); break;
            case 2: go_split<2, FUSE, RNG>(f, grid, nt, a, b, coef, cstride, m, out, ostride, nvec, n, st, ra, gs); break;
            case 3: go_split<3, FUSE, RNG>(f, grid, nt, a, b, coef, cstride, m, out, ostride, nvec, n, st, ra, gs); break;
            case 4: go_split<4, FUSE, RNG>(f, grid, nt, a, b, coef, cstride, m, out, ostride, nvec, n, st, ra, gs); break;
            default: return 1;
        }
        if (RNG && t > 0 && ra.dev_key && !ra.release && !ra.no_advance)
            hipLaunchKernelGGL((k_rng_advance<0>), dim3(1), dim3(1), 0, st, const_cast<RngKey*>(ra.dev_key), 1u);
        return 0;
    }
    // fused chain gate: both factors given as recombinations (GateSrc), product re-shared with the device CSPRNG
    static int gate(const void* Fp, int device, const void* const* rowsA, const uint64_t* lamA2, int kA,
                    const void* const* rowsB, const uint64_t* lamB2, int kB, int t, int m, void* out, size_t ostride,
                    size_t n, hipStream_t st, const RngArgs* rng, int nbatch, size_t yA, size_t yB, size_t yO) {
        const F& f = *reinterpret_cast<const F*>(Fp);
        if (t < 1 || t > 3 || kA < 1 || kA > GATE_MAXK || kB < 0 || kB > GATE_MAXK || !rng) return 2;
        if (nbatch < 1 || nbatch > 255) return 2;
        LaunchCfg lc = launch_cfg(device);
        GateSrc<F> gs;
        memset(&gs, 0, sizeof(gs));
        bool vec = al(out) && (stride_ok(ostride) || m <= 1);
        if (nbatch > 1) {
            gs.yA = yA;
            gs.yB = yB;
            gs.yO = yO;
            vec = vec && stride_ok(yA) && stride_ok(yO) && (kB == 0 || stride_ok(yB));
        }
        for (int j = 0; j < kA; ++j) {
            gs.rowsA[j] = (const E*)rowsA[j];
            gs.lamA[j] = f.prep(word_at<F>(f, lamA2, j));
            vec = vec && al(rowsA[j]);
        }
        for (int j = 0; j < kB; ++j) {
            gs.rowsB[j] = (const E*)rowsB[j];
            gs.lamB[j] = f.prep(word_at<F>(f, lamB2, j));
            vec = vec && al(rowsB[j]);
        }
        gs.kA = kA;
        gs.kB = kB;
        gs.square = kB == 0;
        constexpr int SLG = scalar_limbs<F>();
        gs.plainA = kA == 1 && lamA2[0] == 1 && lamA2[1] == 0 && (SLG < 3 || lamA2[SLG - 1] == 0);
        gs.plainB = kB == 1 && lamB2[0] == 1 && lamB2[1] == 0 && (SLG < 3 || lamB2[SLG - 1] == 0);
        size_t nvec = nvec_of(n, vec);
        RngArgs ra = *rng;
        const bool spread = nvec > 0 && nvec < 262144;
        ra.spread = spread ? 1 : 0;
        unsigned grid = grid_for(nvec ? (!spread ? (n / EPV + 2) / 2 : nvec) : n, lc);
        const unsigned gy = (unsigned)nbatch;
        ra.release = !ra.no_advance && (size_t)grid * gy <= RNG_RELEASE_MAX_GRID;
        E* o = (E*)out;
        switch (t) {
            case 1: go_split<1, true, true, true>(f, grid, true, nullptr, nullptr, nullptr, 0, m, o, ostride, nvec, n, st, ra, gs, gy); break;
            case 2: go_split<2, true, true, true>(f, grid, true, nullptr, nullptr, nullptr, 0, m, o, ostride, nvec, n, st, ra, gs, gy); break;
            default: go_split<3, true, true, true>(f, grid, true, nullptr, nullptr, nullptr, 0, m, o, ostride, nvec, n, st, ra, gs, gy); break;
        }
        if (ra.dev_key && !ra.release && !ra.no_advance)
            hipLaunchKernelGGL((k_rng_advance<0>), dim3(1), dim3(1), 0, st, const_cast<RngKey*>(ra.dev_key), 1u);
        FFGPU_CHECK_LAUNCH();
        return 0;
    }
    static int split(const void* Fp, int device, const void* a, const void* b, const void* coef,
                     size_t cstride, int t, int m, void* out, size_t ostride, size_t n, hipStream_t st,
                     const RngArgs* rng) {
        const F& f = *reinterpret_cast<const F*>(Fp);
        LaunchCfg lc = launch_cfg(device);
        RngArgs ra;
        memset(&ra, 0, sizeof(ra));
        int rc;
        if (rng) {
            ra = *rng;
            rc = b ? split_t<true, true>(f, lc, (const E*)a, (const E*)b, nullptr, 0, t, m, (E*)out, ostride, n, st, ra)
                   : split_t<false, true>(f, lc, (const E*)a, nullptr, nullptr, 0, t, m, (E*)out, ostride, n, st, ra);
        } else {
            rc = b ? split_t<true, false>(f, lc, (const E*)a, (const E*)b, (const E*)coef, cstride, t, m, (E*)out,
                                          ostride, n, st, ra)
                   : split_t<false, false>(f, lc, (const E*)a, nullptr, (const E*)coef, cstride, t, m, (E*)out,
                                           ostride, n, st, ra);
        }
        if (rc) return rc;
        FFGPU_CHECK_LAUNCH();
        return 0;
    }
    static int rng_coeffs(const void* Fp, int device, void* coef, size_t cstride, int t, size_t n, hipStream_t st,
                          const RngArgs* rng) {
        const F& f = *reinterpret_cast<const F*>(Fp);
        LaunchCfg lc = launch_cfg(device);
        E* C = (E*)coef;
        size_t npacks = (n + EPV - 1) / EPV;
        unsigned grid = grid_for(npacks, lc);
        if (t > MAXT) {
            hipLaunchKernelGGL((k_rng_coeffs_any<F>), dim3(grid), dim3(BLOCK), 0, st, f, C, cstride, t, n, *rng);
            FFGPU_CHECK_LAUNCH();
            return 0;
        }
        bool vec = al(coef) && (stride_ok(cstride) || t <= 1);
        size_t nvec = nvec_of(n, vec);
        switch (t) {
            case 1: hipLaunchKernelGGL((k_rng_coeffs<F, 1>), dim3(grid), dim3(BLOCK), 0, st, f, C, cstride, nvec, n, *rng); break;
            case 2: hipLaunchKernelGGL((k_rng_coeffs<F, 2>), dim3(grid), dim3(BLOCK), 0, st, f, C, cstride, nvec, n, *rng); break;
            case 3: hipLaunchKernelGGL((k_rng_coeffs<F, 3>), dim3(grid), dim3(BLOCK), 0, st, f, C, cstride, nvec, n, *rng); break;
            case 4: hipLaunchKernelGGL((k_rng_coeffs<F, 4>), dim3(grid), dim3(BLOCK), 0, st, f, C, cstride, nvec, n, *rng); break;
            default: return 1;
        }
        FFGPU_CHECK_LAUNCH();
        return 0;
    }

    template <int K>
    static void go_rec(const F& f, const LaunchCfg& lc, const void* const* rows, const uint64_t* lam2, int w,
                       E* out, size_t ostride, size_t n, hipStream_t st) {
        RecArgs<F, K> ra;
        bool vec = al(out) && (stride_ok(ostride) || w <= 1);
        for (int j = 0; j < K; ++j) {
            ra.rows[j] = (const E*)rows[j];
            vec = vec && al(rows[j]);
        }
        for (int r = 0; r < w; ++r)
            for (int j = 0; j < K; ++j) {
                ra.lam[r * K + j] = f.prep(word_at<F>(f, lam2, (size_t)r * K + j));
            }
        for (int i = w * K; i < MAXW * K; ++i) ra.lam[i] = ra.lam[0];
        size_t nvec = nvec_of(n, vec);
        unsigned grid = grid_for(nvec ? nvec : n, lc);
        hipLaunchKernelGGL((k_recombine<F, K, true>), dim3(grid), dim3(BLOCK), 0, st, f, ra, w, out, ostride,
                           nvec, n);
    }
    static int recombine(const void* Fp, int device, const void* const* rows, const uint64_t* lam2, int k,
                         int w, void* out, size_t ostride, size_t n, hipStream_t st) {
        const F& f = *reinterpret_cast<const F*>(Fp);
        LaunchCfg lc = launch_cfg(device);
        E* O = (E*)out;
        if (k > MAXK) {
            if (k > MAXK_ANY) return 2;
            for (int r = 0; r < w; ++r) {
                RecArgsAny<F> ra;
                for (int j = 0; j < k; ++j) {
                    ra.rows[j] = (const E*)rows[j];
                    ra.lam[j] = f.prep(word_at<F>(f, lam2, (size_t)r * k + j));
                }
                for (int j = k; j < MAXK_ANY; ++j) {
                    ra.rows[j] = ra.rows[0];
                    ra.lam[j] = ra.lam[0];
                }
                unsigned grid = grid_for(n, lc);
                hipLaunchKernelGGL((k_recombine_any<F>), dim3(grid), dim3(BLOCK), 0, st, f, ra, k,
                                   O + (size_t)r * ostride, n);
            }
            FFGPU_CHECK_LAUNCH();
            return 0;
        }
        for (int r0 = 0; r0 < w; r0 += MAXW) {
            int wc = (w - r0) < MAXW ? (w - r0) : MAXW;
            const uint64_t* l = lam2 + scalar_limbs<F>() * (size_t)r0 * k;
            E* o = O + (size_t)r0 * ostride;
            switch (k) {
                case 1: go_rec<1>(f, lc, rows, l, wc, o, ostride, n, st); break;
                case 2: go_rec<2>(f, lc, rows, l, wc, o, ostride, n, st); break;
                case 3: go_rec<3>(f, lc, rows, l, wc, o, ostride, n, st); break;
                case 4: go_rec<4>(f, lc, rows, l, wc, o, ostride, n, st); break;
                case 5: go_rec<5>(f, lc, rows, l, wc, o, ostride, n, st); break;
                case 6: go_rec<6>(f, lc, rows, l, wc, o, ostride, n, st); break;
                case 7: go_rec<7>(f, lc, rows, l, wc, o, ostride, n, st); break;
                case 8: go_rec<8>(f, lc, rows, l, wc, o, ostride, n, st); break;
                case 9: go_rec<9>(f, lc, rows, l, wc, o, ostride, n, st); break;
                default: return 1;
            }
        }
        FFGPU_CHECK_LAUNCH();
        return 0;
    }

    static int pow(const void* Fp, int device, const void* a, const ExpArgs* ex, void* out, size_t n,
                   hipStream_t st) {
        const F& f = *reinterpret_cast<const F*>(Fp);
        LaunchCfg lc = launch_cfg(device);
        bool vec = al(a) && al(out);
        size_t nvec = nvec_of(n, vec);
        unsigned grid = grid_for(nvec ? nvec : n, lc);
        hipLaunchKernelGGL((k_pow<F, true>), dim3(grid), dim3(BLOCK), 0, st, f, (const E*)a, *ex, (E*)out, nvec, n);
        FFGPU_CHECK_LAUNCH();
        return 0;
    }
    static int inv(const void* Fp, int device, const void* a, const ExpArgs* ex, void* out, size_t n, int* flag,
                   hipStream_t st) {
        const F& f = *reinterpret_cast<const F*>(Fp);
        LaunchCfg lc = launch_cfg(device);
        bool vec = al(a) && al(out);
        size_t nvec = nvec_of(n, vec);
        // packs per thread: ONE exponentiation (70 products for 2^61 - 1) is shared by G x CH packs, and the
        // G x CH x N prefix words stay in registers (two waves per SIMD at CH = 8..12 for one-word fields)
        if constexpr (F::EPW == 1 && sizeof(W) == 8) {
            // One exponentiation (70 products for 2^61 - 1) is shared by the CH x G packs of a thread: k_inv_fast (full batches
            // without predicates, second reads in a window, exponentiation without its window table when the exponent allows
            // it) at 8 x 2 packs = 32 elements per thread and three waves per SIMD for arrays of at least 32768 elements
            // (profiles/r04_alu.md: thirteen shapes measured; more packs per thread spill or fall to two waves, fewer pay more
            // exponentiations); k_inv_batch below that and for GF(2^n).
            // (prime fields only: the GF(2^n) product has run-time loops, its prefix array lives in scratch memory either way,
            // and the round-3 kernel needs less of it -- 272-336 B against 528-624 B per thread)
            if (!F::BINARY && nvec >= (size_t)BLOCK * 64) {
                if (pow_lean_ok(*ex)) return launch_inv_fast<8, 2, 6, 3, true>(f, a, ex, out, nvec, n, flag, st);
                return launch_inv_fast<8, 2, 3, 1, false>(f, a, ex, out, nvec, n, flag, st);
            }
            // All waves of the launch take the same time and two fit on a SIMD, so the launch runs in ROUNDS of
            // 2 x 4 x num_cu waves: 10^7 elements at CH = 8 are 4883 waves = 2.4 rounds -- three rounds of time for
            // 2.4 of work (measured: 56 us).  More packs per thread amortise the exponentiation better AND change the
            // number of rounds; pick the CH with the least rounds x (products per thread).
            const size_t slots = (size_t)lc.num_cu * 4 * 2;
            int best = 8;
            double best_cost = 0;
            for (int ch : {8, 10}) {                    // (CH = 12: 296 VGPRs, one wave per SIMD)
                const size_t waves = (nvec / (size_t)(ch * 2) + 63) / 64 + 1;
                const size_t rounds = (waves + slots - 1) / slots;
                const double cost = (double)rounds * (3.0 * ch * 2 * (double)EPV + 73.0);
                if (ch == 8 || cost < best_cost * 0.97) {
                    best = ch;
                    best_cost = ch == 8 ? cost : (cost < best_cost ? cost : best_cost);
                }
            }
            if (best == 10) return launch_inv<10, 2>(f, lc, a, ex, out, nvec, n, flag, st);
            return launch_inv<8, 2>(f, lc, a, ex, out, nvec, n, flag, st);
        } else {
            if constexpr (HasDigitChain<F>::value) {
                // multi-limb 2^k - c primes: the whole batch in digits (k_inv_digits), NL = ceil(k / 28) registers per value
                if (nvec >= 4096) {
                    int rc = launch_inv_digits_from<F::CHAIN_MIN_NL>(f, lc, a, ex, out, nvec, n, flag, st);
                    if (rc >= 0) return rc;
                }
            }
            constexpr int CH = F::EPW > 1 ? 2 : 8;          // packed bytes: 8 words per batch (zero mask)
            return launch_inv<CH, 1>(f, lc, a, ex, out, nvec, n, flag, st);
        }
    }
    // -1: the modulus has no digit form at this NL (DigitChain::setup: c too large) -> the word kernel
    template <int NL>
    static int launch_inv_digits_from(const F& f, const LaunchCfg& lc, const void* a, const ExpArgs* ex, void* out, size_t nvec,
                                      size_t n, int* flag, hipStream_t st) {
        if constexpr (!HasDigitChain<F>::value) {
            return -1;
        } else if constexpr (NL > F::CHAIN_MAX_NL) {
            return -1;
        } else {
            if (f.k > 28u * NL) return launch_inv_digits_from<NL + 1>(f, lc, a, ex, out, nvec, n, flag, st);
            DigitChain<NL> dc;
            if (!f.template chain_setup<NL>(dc)) return -1;
            constexpr int CH = NL <= 3 ? 32 : NL == 4 ? 24 : NL <= 6 ? 16 : 12;      // ~96 registers of prefix products
            unsigned grid = grid_for((nvec + CH - 1) / CH, lc);
            hipLaunchKernelGGL((k_inv_digits<F, NL, CH, true>), dim3(grid), dim3(BLOCK), 0, st, f, (const E*)a, *ex, (E*)out, nvec, n,
                               flag);
            FFGPU_CHECK_LAUNCH();
            return 0;
        }
    }
    // square-and-multiply after the leading run costs popcount(tail) products, the window table 8 up front and one per
    // window: lean when the tail holds few set bits (q - 2 of every 2^k - c prime: a run of ones and a short tail)
    static bool pow_lean_ok(const ExpArgs& ex) {
        int i = ex.nbits - 1;
        auto bit = [&](int b) { return (int)((ex.e[b >> 6] >> (b & 63)) & 1u); };
        while (i >= 0 && bit(i)) --i;                 // the leading run
        if (ex.nbits - 1 - i < 12) i = ex.nbits - 2;  // (short runs are not raised by doubling: everything is tail)
        int ones = 0;
        for (int b = i; b >= 0; --b) ones += bit(b);
        return ones <= 6;
    }
    template <int CH, int G, int WIN, int WAVES, bool LEAN, int WIN1 = 0>
    static int launch_inv_fast(const F& f, const void* a, const ExpArgs* ex, void* out, size_t nvec, size_t n, int* flag,
                               hipStream_t st) {
        if constexpr (F::EPW == 1 && sizeof(W) == 8) {
            const size_t per_block = (size_t)BLOCK * CH * G;
            const size_t nfull = nvec / per_block;
            const size_t rest = nvec % per_block;
            const unsigned grid = (unsigned)nfull + (unsigned)((rest + BLOCK - 1) / BLOCK) + ((rest == 0 && n > nvec * EPV) ? 1u : 0u);
            hipLaunchKernelGGL((k_inv_fast<F, CH, G, WIN, WAVES, LEAN, WIN1>), dim3(grid), dim3(BLOCK), 0, st, f, (const E*)a, *ex, (E*)out,
                               nvec, n, (unsigned)nfull, flag);
            FFGPU_CHECK_LAUNCH();
        }
        return 0;
    }
    template <int CH, int G>
    static int launch_inv(const F& f, const LaunchCfg& lc, const void* a, const ExpArgs* ex, void* out, size_t nvec, size_t n,
                          int* flag, hipStream_t st) {
        size_t iters = nvec ? (nvec + CH * G - 1) / (CH * G) : n;
        unsigned grid = grid_for(iters, lc);
        hipLaunchKernelGGL((k_inv_batch<F, CH, G, true>), dim3(grid), dim3(BLOCK), 0, st, f, (const E*)a, *ex, (E*)out,
                           nvec, n, flag);
        FFGPU_CHECK_LAUNCH();
        return 0;
    }

    // skinny shapes (one output dimension <= 8): HBM-bound kernels that read the big operand once
    template <int NN>
    static void go_matvec(const F& f, const E* A, size_t lda, const E* B, size_t ldb, E* C, size_t ldc, int M, int K,
                          int N, hipStream_t st) {
        if (K <= 32 && M >= 1024) {               // short rows: one thread per row
            unsigned grid = (unsigned)(((size_t)M + BLOCK - 1) / BLOCK);
            hipLaunchKernelGGL((k_matvec_short_rows<F, NN>), dim3(grid), dim3(BLOCK), 0, st, f, A, lda, B, ldb, C, ldc, M, K, N);
            return;
        }
        const int vec = al(A) && stride_ok(lda);
        const int bvec = al(B) && stride_ok(ldb) && (N % (int)Pack<W>::N == 0) && sizeof(E) != 12;
        if constexpr (NN <= 2) {              // (12-byte elements too since round 6: one element per pack, dwordx3 accesses)
            // long rows, one or two columns: R = 2 rows per workgroup share every load of B (measured at 4096^2 / 8192^2:
            // R = 1 32.5 / 117 us, R = 2 27.4 / 85 us, R = 4 28.3 / 99 us, R = 8 35.8 / 112 us)
            constexpr int R = 2;
            if (vec && K >= 1024 && M >= 1024 * R) {
                const int bpack = (N == 1 && ldb == 1 && al(B)) ? 1 : 0;
                hipLaunchKernelGGL((k_matvec_rows_r<F, NN, R>), dim3((unsigned)((M + R - 1) / R)), dim3(BLOCK), 0, st, f, A, lda, B,
                                   ldb, C, ldc, M, K, N, vec, bpack);
                return;
            }
        }
        hipLaunchKernelGGL((k_matvec_rows<F, NN>), dim3((unsigned)M), dim3(BLOCK), 0, st, f, A, lda, B, ldb, C, ldc, K, N, vec,
                           bvec);
    }
    // one-word primes, 2..8 columns: column sums, one instantiation per N; two rows of A per workgroup up to 4 columns
    template <int NN>
    static void go_matvec_col(const F& f, int device, const E* A, size_t lda, const E* B, size_t ldb, E* C, size_t ldc, int M,
                              int K, hipStream_t st) {
        if constexpr (col_mac_ok<F>::value) {
            constexpr int R = NN <= 4 ? 2 : 1;
            const int vec = al(A) && stride_ok(lda);
            const int bvec = al(B) && stride_ok(ldb) && (NN % (int)Pack<W>::N == 0);
            // B contiguous and 16-byte aligned, rows of A aligned: sixteen lanes per row (a wave per row when 16 rows per
            // workgroup would leave CUs without one), the rows of B through LDS
            if (vec && ldb == (size_t)NN && al(B) && K >= 64) {
                const int ncu = launch_cfg(device).num_cu;
                if ((M + 15) / 16 >= 2 * (ncu > 0 ? ncu : 256))
                    hipLaunchKernelGGL((k_matvec_sub_col<F, NN, 16>), dim3((unsigned)((M + 15) / 16)), dim3(BLOCK), 0, st, f, A, lda, B, C, ldc, M, K);
                else
                    hipLaunchKernelGGL((k_matvec_sub_col<F, NN, 64>), dim3((unsigned)((M + 3) / 4)), dim3(BLOCK), 0, st, f, A, lda, B, C, ldc, M, K);
            } else {
                hipLaunchKernelGGL((k_matvec_rows_col<F, NN, R>), dim3((unsigned)((M + R - 1) / R)), dim3(BLOCK), 0, st, f, A, lda, B, ldb,
                                   C, ldc, M, K, vec, bvec);
            }
        }
    }
    template <int MM>
    static void go_vecmat(const F& f, const E* A, size_t lda, const E* B, size_t ldb, W* part, int M, int K, int N,
                          int ks, int kchunk, hipStream_t st) {
        constexpr int CW = Pack<W>::N;
        const bool vec = sizeof(E) != 12 && CW > 1 && al(B) && stride_ok(ldb) && N % CW == 0;
        if (vec) {
            dim3 grid((N / CW + BLOCK - 1) / BLOCK, ks);
            hipLaunchKernelGGL((k_vecmat_partial<F, MM, true, (MM > 1)>), grid, dim3(BLOCK), 0, st, f, A, lda, B, ldb, part, M, K, N, kchunk);
        } else {
            dim3 grid((N + BLOCK - 1) / BLOCK, ks);
            hipLaunchKernelGGL((k_vecmat_partial<F, MM, false, (MM > 1)>), grid, dim3(BLOCK), 0, st, f, A, lda, B, ldb, part, M, K, N, kchunk);
        }
    }
    // one-word prime fields: column accumulators, one instantiation per M, one column per thread.  Rows of B per group (two
    // groups in registers) and resident workgroups per CU follow the registers of the column sums, 12 per row of A (measured,
    // 4096 x 4096 over 2^61 - 1, profiles/HISTORY.md): up to 4 rows four workgroups and groups of 8; 5..7 rows three
    // workgroups and groups of 4; 8 rows two workgroups and groups of 16
    template <int MM>
    struct VecmatCol {
        static constexpr int UNR = MM <= 4 ? 8 : MM < 8 ? 4 : 16;
        static constexpr int MINB = MM <= 4 ? 4 : MM < 8 ? 3 : 2;
    };
    static int vecmat_col_per_cu(int M) { return M <= 4 ? 4 : M < 8 ? 3 : 2; }
    template <int MM>
    static void go_vecmat_col(const F& f, const E* A, size_t lda, const E* B, size_t ldb, W* part, int K, int N, int ks,
                              int kchunk, hipStream_t st) {
        if constexpr (col_mac_ok<F>::value) {
            typedef VecmatCol<MM> C;
            dim3 grid((N + BLOCK - 1) / BLOCK, ks);
            hipLaunchKernelGGL((k_vecmat_partial_col<F, MM, C::UNR, C::MINB>), grid, dim3(BLOCK), 0, st, f, A, lda, B, ldb, part, K, N, kchunk);
        }
    }
    static int matmul(const void* Fp, int device, const void* A, size_t lda, const void* B, size_t ldb, void* C,
                      size_t ldc, int M, int K, int N, void* workspace, size_t workspace_bytes, const Tuning* tune,
                      hipStream_t st) {
        const F& f = *reinterpret_cast<const F*>(Fp);
        if constexpr (F::EPW == 1) {
            // (three-limb words: the eight-column kernel would spill, N in 5..8 takes the tiled product)
            if (N <= (sizeof(W) > 16 ? 4 : SKINNY_MAX) && M >= 64 && K >= 1) {
                const E* a = (const E*)A; const E* b = (const E*)B; E* c = (E*)C;
                bool done = false;
                if constexpr (col_mac_ok<F>::value) {
                    if (N >= 2 && K > 32) {               // (short rows keep the one-thread-per-row kernel)
                        switch (N) {
                            case 2: go_matvec_col<2>(f, device, a, lda, b, ldb, c, ldc, M, K, st); break;
                            case 3: go_matvec_col<3>(f, device, a, lda, b, ldb, c, ldc, M, K, st); break;
                            case 4: go_matvec_col<4>(f, device, a, lda, b, ldb, c, ldc, M, K, st); break;
                            case 5: go_matvec_col<5>(f, device, a, lda, b, ldb, c, ldc, M, K, st); break;
                            case 6: go_matvec_col<6>(f, device, a, lda, b, ldb, c, ldc, M, K, st); break;
                            case 7: go_matvec_col<7>(f, device, a, lda, b, ldb, c, ldc, M, K, st); break;
                            default: go_matvec_col<8>(f, device, a, lda, b, ldb, c, ldc, M, K, st); break;
                        }
                        done = true;
                    }
                }
                if (done) {}
                else if (N == 1) go_matvec<1>(f, a, lda, b, ldb, c, ldc, M, K, N, st);
                else if (N == 2) go_matvec<2>(f, a, lda, b, ldb, c, ldc, M, K, N, st);
                else if (N <= 4) go_matvec<4>(f, a, lda, b, ldb, c, ldc, M, K, N, st);
                else go_matvec<8>(f, a, lda, b, ldb, c, ldc, M, K, N, st);
                FFGPU_CHECK_LAUNCH();
                return 0;
            }
            if (M <= SKINNY_MAX && N >= 64 && K >= 1 && workspace) {
                // split K so that one round of workgroups fills the chip; each chunk at least 8 rows.  One-word primes: one
                // column per thread, as many workgroups as are resident at once (registers of the column sums)
                int cpt = (int)Pack<W>::N, target = 1024;        // columns per thread, workgroups
                if constexpr (col_mac_ok<F>::value) {
                    cpt = 1;
                    const int ncu = launch_cfg(device).num_cu;
                    target = (ncu > 0 ? ncu : 256) * vecmat_col_per_cu(M);
                }
                const int cols_blocks = (N / cpt + BLOCK - 1) / BLOCK;
                int ks = (target + cols_blocks - 1) / cols_blocks;
                if (ks > (K + 7) / 8) ks = (K + 7) / 8;
                if (ks < 1) ks = 1;
                while (ks > 1 && (size_t)ks * M * N * sizeof(W) > workspace_bytes) ks /= 2;
                if ((size_t)ks * M * N * sizeof(W) <= workspace_bytes) {
                    const int kchunk = (K + ks - 1) / ks;
                    ks = (K + kchunk - 1) / kchunk;
                    W* part = (W*)workspace;
                    const E* a = (const E*)A; const E* b = (const E*)B;
                    if constexpr (col_mac_ok<F>::value) {
                        switch (M) {
                            case 1: go_vecmat_col<1>(f, a, lda, b, ldb, part, K, N, ks, kchunk, st); break;
                            case 2: go_vecmat_col<2>(f, a, lda, b, ldb, part, K, N, ks, kchunk, st); break;
                            case 3: go_vecmat_col<3>(f, a, lda, b, ldb, part, K, N, ks, kchunk, st); break;
                            case 4: go_vecmat_col<4>(f, a, lda, b, ldb, part, K, N, ks, kchunk, st); break;
                            case 5: go_vecmat_col<5>(f, a, lda, b, ldb, part, K, N, ks, kchunk, st); break;
                            case 6: go_vecmat_col<6>(f, a, lda, b, ldb, part, K, N, ks, kchunk, st); break;
                            case 7: go_vecmat_col<7>(f, a, lda, b, ldb, part, K, N, ks, kchunk, st); break;
                            default: go_vecmat_col<8>(f, a, lda, b, ldb, part, K, N, ks, kchunk, st); break;
                        }
                    } else if (M == 1) go_vecmat<1>(f, a, lda, b, ldb, part, M, K, N, ks, kchunk, st);
                    else if (M == 2) go_vecmat<2>(f, a, lda, b, ldb, part, M, K, N, ks, kchunk, st);
                    else if (M <= 4) go_vecmat<4>(f, a, lda, b, ldb, part, M, K, N, ks, kchunk, st);
                    else go_vecmat<8>(f, a, lda, b, ldb, part, M, K, N, ks, kchunk, st);
                    constexpr int CO = BLOCK / VECMAT_FINAL_G;
                    hipLaunchKernelGGL((k_vecmat_final<F>), dim3((unsigned)(((size_t)M * N + CO - 1) / CO)), dim3(BLOCK), 0,
                                       st, f, (const W*)part, ks, M, N, (E*)C, ldc);
                    FFGPU_CHECK_LAUNCH();
                    return 0;
                }
            }
        }
        const bool use_mfma = !tune || tune->mm_mfma != 0;
        const double mfma_min = tune ? tune->mm_mfma_min : 8e7;
        if constexpr (F::EPW == 1 && !F::BINARY && sizeof(W) <= 8) {
            // large dense products over primes of up to 64 bits: int8 matrix cores, 8 signed base-256 digits per operand
            // (k_limb_gemm_glds), 4 for 32-bit storage (k_limb_gemm_l4)
            const int L = sizeof(W) == 4 ? 4 : 8;
            const uint64_t pmod = (uint64_t)f.p;
            const int Mp = (M + 63) / 64 * 64, Np = (N + 63) / 64 * 64, Kp = (K + 31) / 32 * 32;
            const size_t need = (size_t)L * ((size_t)Mp + Np) * Kp;
            // (from 9 rows / columns on: the tiles are padded to 64 -- a batch of 9..63 rows against 4096 x 4096 takes the 88 us of
            // 64 rows instead of 670-690 us on the vector ALUs)
            if (use_mfma && M > SKINNY_MAX && N > SKINNY_MAX && K >= 64 && (double)M * N * K >= mfma_min && workspace && need <= workspace_bytes) {
                int8_t* Ap = (int8_t*)workspace;
                int8_t* Bp = Ap + (size_t)L * Mp * Kp;
                const unsigned ga = (unsigned)(((size_t)Mp * Kp + BLOCK - 1) / BLOCK);
                dim3 gb(Np / 32, Kp / 32), gg(Np / 64, Mp / 64);
                auto go = [&](auto lc_) {
                    constexpr int LL = decltype(lc_)::value;
                    hipLaunchKernelGGL((k_limb_split_a<F, LL>), dim3(ga), dim3(BLOCK), 0, st, (const E*)A, lda, pmod, Ap, M, K, Mp, Kp);
                    // up to 128 rows of 64-bit elements: the product kernel converts B itself (BRAW), no digit planes of B --
                    // for whole tiles and 16-byte aligned rows of B; ragged shapes go through the planes
                    constexpr bool CAN_RAW = LL == 8 && sizeof(E) == 8;
                    const bool braw = CAN_RAW && gg.y <= 2 && K % 32 == 0 && N % 64 == 0 && ldb % 2 == 0 && (((uintptr_t)B) & 15) == 0;
                    if (!braw)
                        hipLaunchKernelGGL((k_limb_split_bt<F, LL>), gb, dim3(BLOCK), 0, st, (const E*)B, ldb, pmod, Bp, K, N, Np, Kp);
                    auto product = [&](dim3 grid, E* out, size_t out_ld, int kb, int ke, int acc_, int kslice, size_t zs) {
                        if constexpr (CAN_RAW) {
                            if (braw) launch_glds<true>(f, grid, st, Ap, (const int8_t*)nullptr, out, out_ld, M, N, Kp, kb, ke, acc_, kslice, zs, (const E*)B, ldb, pmod);
                            else launch_glds<false>(f, grid, st, Ap, Bp, out, out_ld, M, N, Kp, kb, ke, acc_, kslice, zs, (const E*)nullptr, (size_t)0, (uint64_t)0);
                        } else {
                            hipLaunchKernelGGL((k_limb_gemm_l4<F>), grid, dim3(BLOCK), 0, st, f, (const int8_t*)Ap, (const int8_t*)Bp, out,
                                               out_ld, M, N, Kp, kb, ke, acc_, kslice, zs);
                        }
                    };
                    // few output tiles (a batch of 64..256 rows against a big matrix): split K over blockIdx.z into
                    // slabs behind the planes, summed by k_splitk_sum
                    const size_t tiles = (size_t)gg.x * gg.y;
                    int ks = 1;
                    if (tiles <= 128 && Kp >= 512) {
                        // ONE round of workgroups (a workgroup holds a CU: 512 registers per lane): tiles x slabs ~ CUs.
                        // Measured (round 4, 64 x 4096 x 4096): 256 workgroups 94 us, 384: 127, 512: 107, 768: 117, 1536: 121
                        // -- every extra slab repeats the epilogue and the pipeline fill.
                        const int ncu = launch_cfg(device).num_cu;
                        const int target = ncu > 0 ? ncu : 256;
                        ks = (int)((target + tiles - 1) / tiles);
                        if (ks > Kp / 256) ks = Kp / 256;
                        while (ks > 1 && need + 256 + (size_t)ks * M * N * sizeof(E) > workspace_bytes) --ks;
                    }
                    if (ks > 1 && Kp <= LIMB_KCHUNK) {
                        const int kslice = ((Kp + ks - 1) / ks + 31) / 32 * 32;
                        ks = (Kp + kslice - 1) / kslice;
                        E* slabs = (E*)((char*)workspace + ((need + 255) / 256) * 256);
                        dim3 g3(gg.x, gg.y, ks);
                        product(g3, slabs, (size_t)N, 0, Kp, 0, kslice, (size_t)M * N);
                        hipLaunchKernelGGL((k_splitk_sum<F>), dim3((unsigned)(((size_t)M * N + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, st, f,
                                           (const E*)slabs, ks, M, N, (E*)C, ldc);
                        return;
                    }
                    for (int kb = 0; kb < Kp; kb += LIMB_KCHUNK) {
                        const int ke = kb + LIMB_KCHUNK < Kp ? kb + LIMB_KCHUNK : Kp;
                        product(gg, (E*)C, ldc, kb, ke, kb > 0 ? 1 : 0, 0, (size_t)0);
                    }
                };
                if (L == 4) go(std::integral_constant<int, 4>());
                else go(std::integral_constant<int, 8>());
                FFGPU_CHECK_LAUNCH();
                return 0;
            }
        }
        if constexpr (F::EPW == 1 && !F::BINARY && sizeof(W) == 16) {
            // primes of 65..128 bits: the matrix-core product in passes over the diagonals (k_limb_gemm_wide)
            constexpr int LW = sizeof(E) == 12 ? 12 : 16;
            const int Mp = (M + 63) / 64 * 64, Np = (N + 63) / 64 * 64, Kp = (K + 31) / 32 * 32;
            const size_t need = (size_t)LW * ((size_t)Mp + Np) * Kp;
            if (use_mfma && M > SKINNY_MAX && N > SKINNY_MAX && K >= 64 && (double)M * N * K >= mfma_min && workspace &&
                need <= workspace_bytes) {
                int8_t* Ap = (int8_t*)workspace;
                int8_t* Bp = Ap + (size_t)LW * Mp * Kp;
                const unsigned ga = (unsigned)(((size_t)Mp * Kp + BLOCK - 1) / BLOCK);
                dim3 gb(Np / 32, Kp / 32), gg(Np / 64, Mp / 64);
                hipLaunchKernelGGL((k_limb_split_a_wide<F, LW>), dim3(ga), dim3(BLOCK), 0, st, (const E*)A, lda, f.p_lo, f.p_hi, Ap, M,
                                   K, Mp, Kp);
                hipLaunchKernelGGL((k_limb_split_bt_wide<F, LW>), gb, dim3(BLOCK), 0, st, (const E*)B, ldb, f.p_lo, f.p_hi, Bp, K, N,
                                   Np, Kp);
                // 256^D0 mod p by repeated doubling of the canonical 1 (host, canonical arithmetic of the policy)
                auto pow256 = [&](int d0) {
                    W v;
                    v.lo = 1;
                    v.hi = 0;
                    for (int i = 0; i < 8 * d0; ++i) v = f.add(v, v);
                    return v;
                };
                bool first = true;
                for (int kb = 0; kb < Kp; kb += LIMB_KCHUNK_WIDE) {
                    const int ke = kb + LIMB_KCHUNK_WIDE < Kp ? kb + LIMB_KCHUNK_WIDE : Kp;
                    auto pass = [&](auto d0_, auto ndp_) {
                        constexpr int D0 = decltype(d0_)::value, NDP = decltype(ndp_)::value;
                        hipLaunchKernelGGL((k_limb_gemm_wide<F, LW, D0, NDP>), gg, dim3(BLOCK), 0, st, f, (const int8_t*)Ap,
                                           (const int8_t*)Bp, (E*)C, ldc, M, N, Mp, Np, Kp, kb, ke, first ? 0 : 1, pow256(D0));
                        first = false;
                    };
                    if constexpr (LW == 12) {            // 23 diagonals: 12 + 11
                        pass(std::integral_constant<int, 0>(), std::integral_constant<int, 12>());
                        pass(std::integral_constant<int, 12>(), std::integral_constant<int, 11>());
                    } else {                              // 31 diagonals: 11 + 10 + 10
                        pass(std::integral_constant<int, 0>(), std::integral_constant<int, 11>());
                        pass(std::integral_constant<int, 11>(), std::integral_constant<int, 10>());
                        pass(std::integral_constant<int, 21>(), std::integral_constant<int, 10>());
                    }
                }
                FFGPU_CHECK_LAUNCH();
                return 0;
            }
        }
        if constexpr (F::EPW > 1) {
            dim3 grid((N + 31) / 32, (M + 31) / 32);
            hipLaunchKernelGGL((k_matmul_bytes<F>), grid, dim3(BLOCK), 0, st, f, (const uint8_t*)A, lda,
                               (const uint8_t*)B, ldb, (uint8_t*)C, ldc, M, K, N);
        } else {
            // 4 x 2 outputs per thread: measured best (1.93 T MAC/s at 4096^3 over GF(2^61-1))
            // small outputs (a 64 x 64 product is two 64 x 32 tiles): 32 x 32 tiles give four times as many workgroups
            const bool small_out = ((M + 63) / 64) * ((N + 31) / 32) < 64;
            const int tcode = (sizeof(W) >= 16 || small_out) ? 22 : 42;   // two- and three-limb words: 2x2 keeps two waves per SIMD
            const int bm = tcode == 42 ? 64 : 32;
            const int bn = 32;
            dim3 grid((N + bn - 1) / bn, (M + bm - 1) / bm);
            // too few output tiles to fill 256 CUs: split K over blockIdx.z into slabs of the workspace
            int ks = 1, kchunk = 0;
            const size_t tiles = (size_t)grid.x * grid.y;
            E* out = (E*)C;
            size_t out_ld = ldc, zstride = 0;
            if (tiles < 512 && K >= 64 && workspace) {
                ks = (int)((1024 + tiles - 1) / tiles);
                if (ks > K / 32) ks = K / 32;
                while (ks > 1 && (size_t)ks * M * N * sizeof(E) > workspace_bytes) ks /= 2;
                if (ks > 1) {
                    kchunk = ((K + ks - 1) / ks + 15) / 16 * 16;
                    ks = (K + kchunk - 1) / kchunk;
                    grid.z = ks;
                    out = (E*)workspace;
                    out_ld = N;
                    zstride = (size_t)M * N;
                }
            }
            if (ks <= 1) kchunk = 0;
            const E* a = (const E*)A; const E* b = (const E*)B;
            if (tcode == 42)
                hipLaunchKernelGGL((k_matmul<F, 4, 2>), grid, dim3(BLOCK), 0, st, f, a, lda, b, ldb, out, out_ld, M, K, N, kchunk, zstride);
            else if (tcode == 22)
                hipLaunchKernelGGL((k_matmul<F, 2, 2>), grid, dim3(BLOCK), 0, st, f, a, lda, b, ldb, out, out_ld, M, K, N, kchunk, zstride);
            else if (tcode == 84)
                hipLaunchKernelGGL((k_matmul<F, 8, 4>), grid, dim3(BLOCK), 0, st, f, a, lda, b, ldb, out, out_ld, M, K, N, kchunk, zstride);
            else
                hipLaunchKernelGGL((k_matmul<F, 4, 4>), grid, dim3(BLOCK), 0, st, f, a, lda, b, ldb, out, out_ld, M, K, N, kchunk, zstride);
            if (ks > 1)
                hipLaunchKernelGGL((k_splitk_sum<F>), dim3((unsigned)(((size_t)M * N + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, st, f,
                                   (const E*)workspace, ks, M, N, (E*)C, ldc);
        }
        FFGPU_CHECK_LAUNCH();
        return 0;
    }
    static int dot(const void* Fp, int device, const void* a, const void* b, void* out, void* workspace, size_t n,
                   hipStream_t st) {
        const F& f = *reinterpret_cast<const F*>(Fp);
        LaunchCfg lc = launch_cfg(device);
        bool vec = al(a) && (!b || al(b));
        size_t nvec = nvec_of(n, vec);
        size_t iters = nvec ? nvec : n;
        size_t want = (iters + (size_t)BLOCK * 8 - 1) / ((size_t)BLOCK * 8);     // >= 8 packs per thread
        unsigned grid = (unsigned)(want < 1 ? 1 : want > DOT_MAX_BLOCKS ? DOT_MAX_BLOCKS : want);
        W* part = (W*)workspace;
        if (b)
            hipLaunchKernelGGL((k_dot_partial<F, true>), dim3(grid), dim3(BLOCK), 0, st, f, (const E*)a, (const E*)b,
                               part, nvec, n);
        else
            hipLaunchKernelGGL((k_dot_partial<F, false>), dim3(grid), dim3(BLOCK), 0, st, f, (const E*)a, (const E*)a,
                               part, nvec, n);
        hipLaunchKernelGGL((k_dot_final<F>), dim3(1), dim3(BLOCK), 0, st, f, (const W*)part, (int)grid, (E*)out);
        FFGPU_CHECK_LAUNCH();
        (void)lc;
        return 0;
    }
    static int sqrt_cl(const void* Fp, int device, const void* a, const ExpArgs* eleg, const ExpArgs* elad, void* out,
                       size_t n, hipStream_t st) {
        if constexpr (F::BINARY) {
            return 2;
        } else {
            const F& f = *reinterpret_cast<const F*>(Fp);
            LaunchCfg lc = launch_cfg(device);
            unsigned grid = grid_for(n, lc);
            hipLaunchKernelGGL((k_sqrt_cl<F>), dim3(grid), dim3(BLOCK), 0, st, f, (const E*)a, *eleg, *elad, (E*)out, n);
            FFGPU_CHECK_LAUNCH();
            return 0;
        }
    }
    static int gauss(const void* Fp, int device, void* A, int n, int ncols, size_t batch, int det_mode,
                     const ExpArgs* ex, void* det, int* sing, hipStream_t st) {
        const F& f = *reinterpret_cast<const F*>(Fp);
        (void)device;
        constexpr int TI = 4;
        constexpr size_t ZMAX = 32768;                    // grid.z limit: larger batches go in chunks
        for (size_t b0 = 0; b0 < batch; b0 += ZMAX) {
            unsigned nb = (unsigned)(batch - b0 < ZMAX ? batch - b0 : ZMAX);
            E* Ab = (E*)A + b0 * (size_t)n * ncols;
            E* db = det ? (E*)det + b0 : nullptr;
            for (int k = 0; k < n; ++k) {
                hipLaunchKernelGGL((k_gauss_pivot<F>), dim3(nb), dim3(BLOCK), 0, st, f, Ab, n, ncols, k, *ex, db,
                                   sing + b0);
                int cols = ncols - k - 1;
                int rows = det_mode ? n - k - 1 : n;
                if (cols > 0 && rows > 0) {
                    dim3 grid((cols + BLOCK - 1) / BLOCK, (rows + TI - 1) / TI, nb);
                    hipLaunchKernelGGL((k_gauss_elim<F, TI>), grid, dim3(BLOCK), 0, st, f, Ab, n, ncols, k, det_mode,
                                       sing + b0);
                }
            }
        }
        FFGPU_CHECK_LAUNCH();
        return 0;
    }
    static int group_matvec(const void* Fp, int device, const uint64_t* m2, const uint64_t* bias2, int r, int g,
                            const void* in, void* out, size_t ngroups, hipStream_t st) {
        const F& f = *reinterpret_cast<const F*>(Fp);
        if (r < 1 || g < 1 || r > GM_MAX || g > GM_MAX) return 2;
        LaunchCfg lc = launch_cfg(device);
        GroupMatArgs<F> ga;
        memset(&ga, 0, sizeof(ga));
        for (int i = 0; i < r * g; ++i) ga.m[i] = f.prep(word_at<F>(f, m2, i));
        for (int a = 0; a < r; ++a) {
            W b = bias2 ? word_at<F>(f, bias2, a) : word_from_limbs<F>(f, 0, 0);
            if constexpr (F::EPW > 1) b &= 0xffu;     // one element per word on this (element-wise) path
            ga.bias[a] = b;
        }
        ga.r = r;
        ga.g = g;
        unsigned grid = grid_for(ngroups, lc);
        if constexpr (F::EPW == 4) {
            const bool al8 = (((uintptr_t)in) & 7u) == 0;
            if (g == 8 && r == 8 && al8 && (((uintptr_t)out) & 7u) == 0) {
                hipLaunchKernelGGL((k_group8_bytes<F, 8>), dim3(grid), dim3(BLOCK), 0, st, f, ga, (const uint8_t*)in,
                                   (uint8_t*)out, ngroups);
                FFGPU_CHECK_LAUNCH();
                return 0;
            }
            if (g == 8 && r == 1 && al8) {
                hipLaunchKernelGGL((k_group8_bytes<F, 1>), dim3(grid), dim3(BLOCK), 0, st, f, ga, (const uint8_t*)in,
                                   (uint8_t*)out, ngroups);
                FFGPU_CHECK_LAUNCH();
                return 0;
            }
        }
        hipLaunchKernelGGL((k_group_matvec<F>), dim3(grid), dim3(BLOCK), 0, st, f, ga, (const E*)in, (E*)out, ngroups);
        FFGPU_CHECK_LAUNCH();
        return 0;
    }
    static int beaver(const void* Fp, int device, const void* z, const void* x, const void* y, const void* d,
                      const void* e, void* out, int add_de, size_t n, hipStream_t st) {
        const F& f = *reinterpret_cast<const F*>(Fp);
        LaunchCfg lc = launch_cfg(device);
        bool vec = al(z) && al(x) && al(y) && al(d) && al(e) && al(out);
        size_t nvec = nvec_of(n, vec);
        unsigned grid = grid_for(nvec ? nvec : n, lc);
        hipLaunchKernelGGL((k_beaver<F, true>), dim3(grid), dim3(BLOCK), 0, st, f, (const E*)z, (const E*)x, (const E*)y,
                           (const E*)d, (const E*)e, (E*)out, add_de, nvec, n);
        FFGPU_CHECK_LAUNCH();
        return 0;
    }
    static int prss(const void* Fp, int device, const void* const* streams, int ks, int d, int l, int mask_bits,
                    const uint64_t* weights2, const uint64_t* r2, int accumulate, void* out, size_t n,
                    hipStream_t st) {
        const F& f = *reinterpret_cast<const F*>(Fp);
        LaunchCfg lc = launch_cfg(device);
        if (ks < 1 || d < 1 || l < 1 || ks > PRSS_MAXS || ks * d > PRSS_MAXW) return 2;
        PrssArgs<F> pa;
        memset(&pa, 0, sizeof(pa));
        for (int s = 0; s < ks; ++s) pa.streams[s] = (const uint8_t*)streams[s];
        for (int i = 0; i < ks * d; ++i)
            pa.w[i] = f.prep(word_at<F>(f, weights2, i));
        pa.r0 = r2[0];
        pa.r1 = r2[1];
        pa.ks = ks; pa.d = d; pa.l = l; pa.mask_bits = mask_bits; pa.accumulate = accumulate;
        unsigned grid = grid_for(n, lc);
        hipLaunchKernelGGL((k_prss<F>), dim3(grid), dim3(BLOCK), 0, st, f, pa, (E*)out, n);
        FFGPU_CHECK_LAUNCH();
        return 0;
    }

    static int prss_chacha(const void* Fp, int device, const uint8_t* keys40, int ks, int d, int l, int mask_bits, int rounds,
                           const uint64_t* weights2, const uint64_t* r2, int accumulate, void* out, size_t n, hipStream_t st) {
        const F& f = *reinterpret_cast<const F*>(Fp);
        if (ks < 1 || d < 1 || l < 1 || l > 64 || ks > PRSS_CC_MAXS || ks * d > PRSS_CC_MAXW) return 2;
        PrssCcArgs<F> pa;
        memset(&pa, 0, sizeof(pa));
        for (int s = 0; s < ks; ++s) {
            memcpy(pa.key[s], keys40 + 40 * s, 32);
            memcpy(pa.nonce[s], keys40 + 40 * s + 32, 8);
        }
        for (int i = 0; i < ks * d; ++i) pa.w[i] = f.prep(word_at<F>(f, weights2, i));
        pa.r0 = r2[0];
        pa.r1 = r2[1];
        pa.ks = ks; pa.d = d; pa.l = l; pa.mask_bits = mask_bits; pa.accumulate = accumulate; pa.rounds = rounds;
        prss_cc_layout(l, &pa.tb, &pa.dpt);
        const size_t tiles = (n + (size_t)pa.dpt - 1) / (size_t)pa.dpt;
        const unsigned grid = (unsigned)((tiles + BLOCK - 1) / BLOCK);
        hipLaunchKernelGGL((k_prss_chacha<F>), dim3(grid), dim3(BLOCK), 0, st, f, pa, (E*)out, n);
        FFGPU_CHECK_LAUNCH();
        return 0;
    }

    static const FieldOps* table() {
        static const FieldOps ops = {&ew2, &ew1, &muladd, &split, &rng_coeffs, &recombine, &pow, &inv, &matmul, &dot, &gate, &sqrt_cl, &gauss, &group_matvec, &beaver, &prss, &prss_chacha};
        return &ops;
    }
};

}  // namespace ffgpu
