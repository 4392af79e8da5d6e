// misc.hip -- kernels that are not templated on a field policy:
//   * the GF(2^8) S-box layer on local/public bytes (demos/np_aes.py:37-43):
//     a byte->byte map, so the 256-entry table is built once per call on the
//     host (t^254 by the reference's addition chain, runtime.py:1356-1367,
//     then the GF(2) affine map), handed over as a kernel argument, copied to
//     LDS by every workgroup, and 16 bytes per lane stream through 16 look-ups;
//   * the streaming copy used as the achievable-HBM-bandwidth yardstick.
#include <stdlib.h>
#include "kernels.hpp"
#include "bitslice.hpp"

using namespace ffgpu;

struct SboxLut {
    uint8_t v[256];
};

FF_HD uint32_t gf_pow254(const GF2P8& f, uint32_t a) {
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

__global__ __launch_bounds__(BLOCK) void k_sbox(SboxLut tab, const uint8_t* __restrict__ in,
                                                 uint8_t* __restrict__ out, size_t nvec, size_t n) {
    __shared__ uint8_t lut[256];
    lut[threadIdx.x] = tab.v[threadIdx.x];  // BLOCK == 256: one table entry per thread
    __syncthreads();
    const size_t gid = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    const size_t gsz = (size_t)gridDim.x * BLOCK;
    const uint4* __restrict__ iv = reinterpret_cast<const uint4*>(in);
    uint4* __restrict__ ov = reinterpret_cast<uint4*>(out);
    for (size_t i = gid; i < nvec; i += gsz) {
        uint4 v = ldg<true>(iv + i);
        uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            uint32_t x = w[q];
            w[q] = (uint32_t)lut[x & 0xff] | ((uint32_t)lut[(x >> 8) & 0xff] << 8) |
                   ((uint32_t)lut[(x >> 16) & 0xff] << 16) | ((uint32_t)lut[x >> 24] << 24);
        }
        stg<true>(ov + i, make_uint4(w[0], w[1], w[2], w[3]));
    }
    for (size_t e = nvec * 16 + gid; e < n; e += gsz) out[e] = lut[in[e]];
}

// host: the byte -> byte map, built with the same packed GF(2^8) arithmetic the element-wise kernels use:
// t^254 by the reference's addition chain (runtime.py:1356-1367), then the GF(2) affine map.
// (api.hip caches it per context: it only depends on (rows8, b).)
int ffgpu_sbox_build_lut(const void* policy, const uint8_t* rows8, uint8_t b, uint8_t* lut256) {
    const GF2P8& f = *reinterpret_cast<const GF2P8*>(policy);
    for (uint32_t t = 0; t < 256; ++t) {
        uint32_t inv = gf_pow254(f, t) & 0xffu;
        uint32_t y = 0;
        for (int r = 0; r < 8; ++r) y |= (uint32_t)(__builtin_popcount(inv & rows8[r]) & 1) << r;
        lut256[t] = (uint8_t)(y ^ b);
    }
    return 0;
}

int ffgpu_launch_sbox(const uint8_t* lut256, int device, const void* in, void* out, size_t n, hipStream_t st) {
    SboxLut tab;
    memcpy(tab.v, lut256, 256);
    LaunchCfg lc = launch_cfg(device);
    bool vec = aligned16(in) && aligned16(out);
    size_t nvec = vec ? n / 16 : 0;
    unsigned grid = grid_for(nvec ? nvec : n, lc);      // uncapped: the per-workgroup table copy is cheap (measured best)
    hipLaunchKernelGGL(k_sbox, dim3(grid), dim3(BLOCK), 0, st, tab, (const uint8_t*)in, (uint8_t*)out, nvec, n);
    FFGPU_CHECK_LAUNCH();
    return 0;
}

// ---- GF(2^8): public byte -> its 8 bits as field elements, optionally added to bit shares ----------
// out[8 i + j] = ((in[i] >> j) & 1) ^ addend[8 i + j]    (runtime.py:4418-4423: c_bits + r_bits)
// One input byte becomes one 8-byte store; a lane takes 2 input bytes -> 16 bytes out.
__device__ __forceinline__ uint64_t spread_bits(uint32_t v) {
    uint64_t t = ((uint64_t)v * 0x0101010101010101ull) & 0x8040201008040201ull;   // bit j alone in byte j
    return ((t + 0x7f7f7f7f7f7f7f7full) >> 7) & 0x0101010101010101ull;             // -> 0/1 per byte
}
__global__ __launch_bounds__(BLOCK) void k_gf8_to_bits(const uint8_t* __restrict__ in, const uint8_t* __restrict__ addend,
                                                        uint8_t* __restrict__ out, size_t npair, size_t n) {
    const size_t gid = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    const size_t gsz = (size_t)gridDim.x * BLOCK;
    const uint16_t* __restrict__ iv = reinterpret_cast<const uint16_t*>(in);
    const uint4* __restrict__ av = reinterpret_cast<const uint4*>(addend);
    uint4* __restrict__ ov = reinterpret_cast<uint4*>(out);
    for (size_t i = gid; i < npair; i += gsz) {
        const uint32_t v = iv[i];
        uint64_t lo = spread_bits(v & 0xffu), hi = spread_bits(v >> 8);
        if (addend) {
            uint4 a = ldg<true>(av + i);
            lo ^= (uint64_t)a.x | ((uint64_t)a.y << 32);
            hi ^= (uint64_t)a.z | ((uint64_t)a.w << 32);
        }
        stg<true>(ov + i, make_uint4((uint32_t)lo, (uint32_t)(lo >> 32), (uint32_t)hi, (uint32_t)(hi >> 32)));
    }
    for (size_t e = 2 * npair + gid; e < n; e += gsz) {
        uint64_t b = spread_bits(in[e]);
        for (int j = 0; j < 8; ++j) out[8 * e + j] = (uint8_t)((b >> (8 * j)) & 1u) ^ (addend ? addend[8 * e + j] : 0);
    }
}

int ffgpu_launch_gf8_to_bits(int device, const void* in, const void* addend, void* out, size_t n, hipStream_t st) {
    LaunchCfg lc = launch_cfg(device);
    bool vec = (((uintptr_t)in) & 1u) == 0 && (((uintptr_t)out) & 15u) == 0 && (!addend || (((uintptr_t)addend) & 15u) == 0);
    size_t npair = vec ? n / 2 : 0;
    unsigned grid = grid_for(npair ? npair : n, lc);
    hipLaunchKernelGGL(k_gf8_to_bits, dim3(grid), dim3(BLOCK), 0, st, (const uint8_t*)in, (const uint8_t*)addend,
                       (uint8_t*)out, npair, n);
    FFGPU_CHECK_LAUNCH();
    return 0;
}

// ---- GF(2^n<=8): a public 8x8 matrix over every group of 8 bytes, on packed bytes ---------------------
// out_r = bias_r + sum_c M[r][c] (x) x_c for the 8 field elements x_0..x_7 of a group (the bit shares of one
// byte: demos/np_aes.py:40-41), optionally followed by np_from_bits (runtime.py:4475-4484: sum_r 2^r out_r)
// in registers.  The group is one 64-bit value; M is applied diagonal by diagonal:
//     out = XOR_d XOR_b  xtime^b(rot_d(v)) & mask[d][b],   mask[d][b] byte r = 0xff iff bit b of M[r][(r+d)%8]
// (rot_d = rotate right by d bytes, two v_alignbyte; empty diagonals / bit planes are skipped by scalar
// branches).  A 0/1 matrix costs ~6 VALU ops per non-empty diagonal (AES: 5), against ~100 for the
// element-by-element loop -- the difference between 0.9 and 6 TB/s for this 16-bytes-per-group kernel.
struct Gf8Group8Args {
    uint64_t mask[8][8];
    int nplanes[8];
    uint64_t bias;
    int fold;       // 1: store sum_r 2^r out_r (one byte) instead of the 8 bytes
    int n, nfold;   // extension degree; fold rounds needed to bring degree n + 6 below n
    uint32_t nmask;
    int nsh, sh[8]; // bit positions of the modulus without its leading term (x^n = sum x^sh[q])
};

// one group (64 bits = 8 elements) -> 8 output bytes, or the folded byte in the low 8 bits when FOLD.
// GENERAL = false: a 0/1 matrix (bit plane 0 only) -- no inner loops, masks in scalar registers.
template <bool FOLD, bool GENERAL>
__device__ __forceinline__ uint64_t gf8_group8_one(const GF2P8& f, const Gf8Group8Args& ga, uint32_t lo, uint32_t hi) {
    uint32_t alo = (uint32_t)ga.bias, ahi = (uint32_t)(ga.bias >> 32);
#pragma unroll
    for (int d = 0; d < 8; ++d) {
        const int np = ga.nplanes[d];
        if (np == 0) continue;                                              // scalar branch: empty diagonal
        const uint32_t s0 = (d & 4) ? hi : lo, s1 = (d & 4) ? lo : hi;     // rotate right by d bytes
        uint32_t plo = (d & 3) ? __builtin_amdgcn_alignbyte(s1, s0, (uint32_t)(d & 3)) : s0;
        uint32_t phi = (d & 3) ? __builtin_amdgcn_alignbyte(s0, s1, (uint32_t)(d & 3)) : s1;
        alo ^= plo & (uint32_t)ga.mask[d][0];
        ahi ^= phi & (uint32_t)(ga.mask[d][0] >> 32);
        if constexpr (GENERAL) {
            for (int b = 1; b < np; ++b) {                                  // higher bit planes of the constants
                plo = f.xtime(plo);
                phi = f.xtime(phi);
                const uint64_t mk = ga.mask[d][b];
                alo ^= plo & (uint32_t)mk;
                ahi ^= phi & (uint32_t)(mk >> 32);
            }
        }
    }
    if constexpr (!FOLD) {
        return (uint64_t)alo | ((uint64_t)ahi << 32);
    } else {
        // sum_r 2^r (x) out_r: shift byte r up by r bits (an unreduced polynomial of degree < n + 7), then fold
        // the bits above n back with x^n = red (nfold rounds; 2 for the AES polynomial)
        uint32_t u = (alo & 0xffu) ^ ((alo >> 7) & (0xffu << 1)) ^ ((alo >> 14) & (0xffu << 2)) ^
                     ((alo >> 21) & (0xffu << 3)) ^ ((ahi << 4) & (0xffu << 4)) ^ ((ahi >> 3) & (0xffu << 5)) ^
                     ((ahi >> 10) & (0xffu << 6)) ^ ((ahi >> 17) & (0xffu << 7));
        for (int it = 0; it < ga.nfold; ++it) {
            const uint32_t h = u >> ga.n;
            u &= ga.nmask;
            for (int q = 0; q < ga.nsh; ++q) u ^= h << ga.sh[q];
        }
        return u & 0xffu;
    }
}

// A lane takes two 16-byte packs (2 groups each) that lie `half` packs apart, so both loads of a wave are
// fully coalesced 1 KiB requests and each lane has 32 bytes in flight (one group per lane leaves the kernel
// latency-bound at 2.3 TB/s).  Output per pack: 16 bytes, or the 2 folded bytes.
template <bool FOLD, bool GENERAL>
__global__ __launch_bounds__(BLOCK) void k_gf8_group8(GF2P8 f, Gf8Group8Args ga, const uint64_t* __restrict__ in,
                                                       uint8_t* __restrict__ out, size_t npack, size_t ngroups) {
    const size_t gid = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    const size_t gsz = (size_t)gridDim.x * BLOCK;
    const uint4* __restrict__ iv = reinterpret_cast<const uint4*>(in);
    const size_t half = (npack + 1) / 2;
    for (size_t i = gid; i < half; i += gsz) {
        const size_t i2 = i + half;
        const bool two = i2 < npack;
        const uint4 a = ldg<true>(iv + i);
        uint4 b = make_uint4(0, 0, 0, 0);
        if (two) b = ldg<true>(iv + i2);
        const uint64_t r0 = gf8_group8_one<FOLD, GENERAL>(f, ga, a.x, a.y), r1 = gf8_group8_one<FOLD, GENERAL>(f, ga, a.z, a.w);
        const uint64_t r2 = gf8_group8_one<FOLD, GENERAL>(f, ga, b.x, b.y), r3 = gf8_group8_one<FOLD, GENERAL>(f, ga, b.z, b.w);
        if constexpr (FOLD) {
            uint16_t* o16 = reinterpret_cast<uint16_t*>(out);
            o16[i] = (uint16_t)((uint32_t)r0 | ((uint32_t)r1 << 8));
            if (two) o16[i2] = (uint16_t)((uint32_t)r2 | ((uint32_t)r3 << 8));
        } else {
            uint4* ov = reinterpret_cast<uint4*>(out);
            stg<true>(ov + i, make_uint4((uint32_t)r0, (uint32_t)(r0 >> 32), (uint32_t)r1, (uint32_t)(r1 >> 32)));
            if (two) stg<true>(ov + i2, make_uint4((uint32_t)r2, (uint32_t)(r2 >> 32), (uint32_t)r3, (uint32_t)(r3 >> 32)));
        }
    }
    for (size_t g = 2 * npack + gid; g < ngroups; g += gsz) {     // tail / buffers that are only 8-byte aligned
        const uint64_t v = in[g];
        const uint64_t r = gf8_group8_one<FOLD, GENERAL>(f, ga, (uint32_t)v, (uint32_t)(v >> 32));
        if constexpr (FOLD) out[g] = (uint8_t)r;
        else reinterpret_cast<uint64_t*>(out)[g] = r;
    }
}

// matrix: (8, 8) field constants (low byte of each 2-limb scalar; NULL = identity), bias: 8 constants or NULL
static bool gf8_group8_args(const GF2P8& f, const uint64_t* m2, const uint64_t* bias2, int fold, Gf8Group8Args& ga) {
    memset(&ga, 0, sizeof(ga));
    for (int r = 0; r < 8; ++r) {
        for (int c = 0; c < 8; ++c) {
            const uint32_t w = m2 ? (uint32_t)(m2[2 * (r * 8 + c)] & 0xffu) : (uint32_t)(r == c);
            const int d = (c - r + 8) & 7;                   // out_r takes x_{(r+d)%8}
            for (int b = 0; b < 8; ++b)
                if ((w >> b) & 1) {
                    ga.mask[d][b] |= 0xffull << (8 * r);
                    if (ga.nplanes[d] < b + 1) ga.nplanes[d] = b + 1;
                }
        }
        if (bias2) ga.bias |= (uint64_t)(bias2[2 * r] & 0xffu) << (8 * r);
    }
    ga.fold = fold;
    ga.n = (int)f.n;
    const uint32_t red = f.red & 0xffu;
    ga.nmask = (1u << f.n) - 1u;
    for (int q = 0; q < 8; ++q)
        if ((red >> q) & 1) ga.sh[ga.nsh++] = q;
    {   // each round maps degree D >= n to at most D - n + deg(red); start from n + 6
        int dr = red ? 31 - __builtin_clz(red) : 0, D = (int)f.n + 6;
        ga.nfold = 0;
        while (D >= (int)f.n && ga.nfold < 16) {
            D = red ? D - (int)f.n + dr : -1;
            ++ga.nfold;
        }
    }
    bool general = false;
    for (int d = 0; d < 8; ++d) general = general || ga.nplanes[d] > 1;
    return general;
}

int ffgpu_launch_gf8_group8(const void* policy, int device, const uint64_t* m2, const uint64_t* bias2, int fold,
                            const void* in, void* out, size_t ngroups, hipStream_t st) {
    const GF2P8& f = *reinterpret_cast<const GF2P8*>(policy);
    Gf8Group8Args ga;
    const bool general = gf8_group8_args(f, m2, bias2, fold, ga);
    LaunchCfg lc = launch_cfg(device);
    const bool vec = (((uintptr_t)in) & 15u) == 0 && (((uintptr_t)out) & (fold ? 1u : 15u)) == 0;
    const size_t npack = vec ? ngroups / 2 : 0;
    unsigned grid = grid_for(npack ? (npack + 1) / 2 : ngroups, lc);
    const uint64_t* iv = (const uint64_t*)in;
    uint8_t* ov = (uint8_t*)out;
    if (fold && general) hipLaunchKernelGGL((k_gf8_group8<true, true>), dim3(grid), dim3(BLOCK), 0, st, f, ga, iv, ov, npack, ngroups);
    else if (fold) hipLaunchKernelGGL((k_gf8_group8<true, false>), dim3(grid), dim3(BLOCK), 0, st, f, ga, iv, ov, npack, ngroups);
    else if (general) hipLaunchKernelGGL((k_gf8_group8<false, true>), dim3(grid), dim3(BLOCK), 0, st, f, ga, iv, ov, npack, ngroups);
    else hipLaunchKernelGGL((k_gf8_group8<false, false>), dim3(grid), dim3(BLOCK), 0, st, f, ga, iv, ov, npack, ngroups);
    FFGPU_CHECK_LAUNCH();
    return 0;
}

// ---- GF(2^8) secure bit decomposition, fused (runtime.py:4411-4423 + demos/np_aes.py:40-42) --------------
// np_to_bits over a binary field: r_modl = sum_b 2^b r_b; c = open(a + r_modl); bits = bits(c) + r_bits.  With
// all parties of a computation on one GPU the steps before and after the opening are one kernel each:
//
//   k_gf8_mask_open:        c[h] = sum_r coef[r] * rows[r][h]  +  sum_p mu[p] * (sum_b 2^b R_p[8h + b])
//       rows / coef: the shares of `a` of the t+1 opening parties, each possibly still a pending recombination of
//       sub-share rows (coef = mu_p * lambda_s, multiplied on the host); R_p: their shares of the random bits.
//   k_gf8_bits_affine_fold: out_y[h] = sum_r 2^r (M (bits(c[h]) + R_y[8h..8h+7]) + bias)_r   for party y = blockIdx.y
//       the public 8x8 matrix M over the bit shares (np_aes.py:40-41) and np_from_bits (runtime.py:4475-4484).
//
// A lane takes 2 output bytes = 16 bytes of bit shares (one dwordx4, fully coalesced); bytes and rows travel as
// 16-bit accesses (they are 1/8 of the traffic).  Two units per lane, half the array apart, as in k_gf8_group8.
enum { GF8_MO_MAXROWS = 32, GF8_MO_MAXP = 8 };
struct Gf8MaskOpenArgs {
    const uint8_t* rows[GF8_MO_MAXROWS];
    const uint8_t* rbits[GF8_MO_MAXP];
    uint32_t coef[GF8_MO_MAXROWS];
    uint32_t mu[GF8_MO_MAXP];
    int nrows, np;
};
// Both kernels are GF(2)-linear maps from the 8 bit-share bytes of a group to one byte, so they are XORs of
// byte-indexed table entries: out = S[c] ^ T_0[r_0] ^ ... ^ T_7[r_7] (T_b[v] = w_b * v for the public constant w_b of
// column b; S = the same map applied to the bits of the public byte, plus the bias).  The tables (2 KiB + 256 B,
// built on the host with the field's own packed arithmetic) arrive as a kernel argument and are copied to LDS by
// every workgroup; 9 LDS byte reads replace ~60 VALU operations per output byte.
struct Gf8ByteTables {
    uint8_t t[8][256];
    uint8_t s[256];
};

__device__ __forceinline__ void gf8_tables_to_lds(const Gf8ByteTables& tb, uint32_t* lds) {
    const uint32_t* src = reinterpret_cast<const uint32_t*>(&tb);
    for (int i = threadIdx.x; i < (int)(sizeof(Gf8ByteTables) / 4); i += BLOCK) lds[i] = src[i];
    __syncthreads();
}
// XOR_b T_b[byte b of (lo, hi)]
__device__ __forceinline__ uint32_t gf8_tab_group(const uint8_t* lds, uint32_t lo, uint32_t hi) {
    const uint32_t a = lds[0 * 256 + (lo & 0xffu)] ^ lds[1 * 256 + ((lo >> 8) & 0xffu)] ^ lds[2 * 256 + ((lo >> 16) & 0xffu)];
    const uint32_t b = lds[3 * 256 + (lo >> 24)] ^ lds[4 * 256 + (hi & 0xffu)] ^ lds[5 * 256 + ((hi >> 8) & 0xffu)];
    return a ^ b ^ lds[6 * 256 + ((hi >> 16) & 0xffu)] ^ lds[7 * 256 + (hi >> 24)];
}

// two units (byte pairs i and i2, `half` pairs apart) share one 32-bit SWAR word: the field arithmetic costs the
// same for 4 packed bytes as for 2.  Rows are loaded eight at a time before the first multiply.
__device__ __forceinline__ uint32_t gf8_mask_open_word(const GF2P8& f, const uint8_t* lds, const Gf8MaskOpenArgs& a,
                                                        size_t i, size_t i2, bool two) {
    GF2P8::acc acc;
    f.acc_zero(acc);
    uint4 v[GF8_MO_MAXP > 2 ? 2 : GF8_MO_MAXP], v2[2];
    const int np0 = a.np < 2 ? a.np : 2;                   // the first two parties' bit shares go out with the rows
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        v[p] = make_uint4(0, 0, 0, 0);
        v2[p] = make_uint4(0, 0, 0, 0);
        if (p < np0) {
            const uint4* __restrict__ rv = reinterpret_cast<const uint4*>(a.rbits[p]);
            v[p] = ldg<true>(rv + i);
            if (two) v2[p] = ldg<true>(rv + i2);
        }
    }
    for (int r0 = 0; r0 < a.nrows; r0 += 8) {
        uint32_t w[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            w[u] = 0;
            if (r0 + u < a.nrows) {
                const uint16_t* __restrict__ row = reinterpret_cast<const uint16_t*>(a.rows[r0 + u]);
                w[u] = (uint32_t)row[i] | (two ? (uint32_t)row[i2] << 16 : 0u);
            }
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (r0 + u < a.nrows) f.acc_mac(acc, a.coef[r0 + u], w[u]);
    }
    for (int p = 0; p < a.np; ++p) {
        uint4 x, x2 = make_uint4(0, 0, 0, 0);
        if (p < 2) {
            x = p == 0 ? v[0] : v[1];
            x2 = p == 0 ? v2[0] : v2[1];
        } else {
            const uint4* __restrict__ rv = reinterpret_cast<const uint4*>(a.rbits[p]);
            x = ldg<true>(rv + i);
            if (two) x2 = ldg<true>(rv + i2);
        }
        const uint32_t word = gf8_tab_group(lds, x.x, x.y) | (gf8_tab_group(lds, x.z, x.w) << 8) |
                              (gf8_tab_group(lds, x2.x, x2.y) << 16) | (gf8_tab_group(lds, x2.z, x2.w) << 24);
        f.acc_mac(acc, a.mu[p], word);
    }
    return f.acc_reduce(acc);
}

__global__ __launch_bounds__(BLOCK) void k_gf8_mask_open(GF2P8 f, Gf8ByteTables tb, Gf8MaskOpenArgs a, uint8_t* __restrict__ out,
                                                          size_t npair, size_t n) {
    __shared__ uint32_t lds32[sizeof(Gf8ByteTables) / 4];
    gf8_tables_to_lds(tb, lds32);
    const uint8_t* lds = reinterpret_cast<const uint8_t*>(lds32);
    const size_t gid = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    const size_t gsz = (size_t)gridDim.x * BLOCK;
    uint16_t* __restrict__ o16 = reinterpret_cast<uint16_t*>(out);
    const size_t half = (npair + 1) / 2;
    for (size_t i = gid; i < half; i += gsz) {
        const size_t i2 = i + half;
        const bool two = i2 < npair;
        const uint32_t w = gf8_mask_open_word(f, lds, a, i, i2, two);
        o16[i] = (uint16_t)w;
        if (two) o16[i2] = (uint16_t)(w >> 16);
    }
    for (size_t e = 2 * npair + gid; e < n; e += gsz) {            // odd tail / unaligned buffers: one byte at a time
        GF2P8::acc acc;
        f.acc_zero(acc);
        for (int r = 0; r < a.nrows; ++r) f.acc_mac(acc, a.coef[r], (uint32_t)a.rows[r][e]);
        for (int p = 0; p < a.np; ++p) {
            uint32_t fb = 0;
            for (int b = 0; b < 8; ++b) fb ^= lds[b * 256 + a.rbits[p][8 * e + b]];
            f.acc_mac(acc, a.mu[p], fb);
        }
        out[e] = (uint8_t)(f.acc_reduce(acc) & 0xffu);
    }
}

__global__ __launch_bounds__(BLOCK) void k_gf8_bits_affine_fold(Gf8ByteTables tb, const uint8_t* __restrict__ c,
                                                                 const uint8_t* __restrict__ rbits, size_t ybr,
                                                                 uint8_t* __restrict__ out, size_t ybo, size_t npair, size_t n) {
    __shared__ uint32_t lds32[sizeof(Gf8ByteTables) / 4];
    gf8_tables_to_lds(tb, lds32);
    const uint8_t* lds = reinterpret_cast<const uint8_t*>(lds32);
    rbits += (size_t)blockIdx.y * ybr;
    out += (size_t)blockIdx.y * ybo;
    const size_t gid = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    const size_t gsz = (size_t)gridDim.x * BLOCK;
    const uint16_t* __restrict__ c16 = reinterpret_cast<const uint16_t*>(c);
    const uint4* __restrict__ rv = reinterpret_cast<const uint4*>(rbits);
    uint16_t* __restrict__ o16 = reinterpret_cast<uint16_t*>(out);
    const size_t half = (npair + 1) / 2;
    for (size_t i = gid; i < half; i += gsz) {
        const size_t i2 = i + half;
        const bool two = i2 < npair;
        const uint4 r = ldg<true>(rv + i);
        uint4 r2 = make_uint4(0, 0, 0, 0);
        uint32_t v = c16[i], v2 = 0;
        if (two) {
            r2 = ldg<true>(rv + i2);
            v2 = c16[i2];
        }
        const uint32_t b0 = lds[2048 + (v & 0xffu)] ^ gf8_tab_group(lds, r.x, r.y);
        const uint32_t b1 = lds[2048 + (v >> 8)] ^ gf8_tab_group(lds, r.z, r.w);
        o16[i] = (uint16_t)(b0 | (b1 << 8));
        if (two) {
            const uint32_t b2 = lds[2048 + (v2 & 0xffu)] ^ gf8_tab_group(lds, r2.x, r2.y);
            const uint32_t b3 = lds[2048 + (v2 >> 8)] ^ gf8_tab_group(lds, r2.z, r2.w);
            o16[i2] = (uint16_t)(b2 | (b3 << 8));
        }
    }
    for (size_t e = 2 * npair + gid; e < n; e += gsz) {
        uint32_t b = lds[2048 + c[e]];
        for (int j = 0; j < 8; ++j) b ^= lds[j * 256 + rbits[8 * e + j]];
        out[e] = (uint8_t)b;
    }
}

// T_b[v] = w_b * v with w_b = sum_r 2^r * M[r][b] (M = NULL: identity, so w_b = 2^b: np_from_bits alone);
// S[c] = sum_b bit_b(c) * w_b + sum_r 2^r * bias_r
static void gf8_byte_tables(const GF2P8& f, const uint64_t* m2, const uint64_t* bias2, Gf8ByteTables& tb) {
    auto mul1 = [&](uint32_t a, uint32_t b) { return f.mul(a & 0xffu, b & 0xffu) & 0xffu; };
    uint32_t w[8], pw[8], biasf = 0;
    pw[0] = 1;
    for (int r = 1; r < 8; ++r) pw[r] = f.xtime(pw[r - 1]) & 0xffu;        // 2^r in the field
    for (int b = 0; b < 8; ++b) {
        w[b] = 0;
        for (int r = 0; r < 8; ++r) {
            const uint32_t mrc = m2 ? (uint32_t)(m2[2 * (r * 8 + b)] & 0xffu) : (uint32_t)(r == b);
            w[b] ^= mul1(pw[r], mrc);
        }
    }
    if (bias2)
        for (int r = 0; r < 8; ++r) biasf ^= mul1(pw[r], (uint32_t)(bias2[2 * r] & 0xffu));
    for (int b = 0; b < 8; ++b)
        for (uint32_t v = 0; v < 256; ++v) tb.t[b][v] = (uint8_t)mul1(w[b], v);
    for (uint32_t cv = 0; cv < 256; ++cv) {
        uint32_t sv = biasf;
        for (int b = 0; b < 8; ++b)
            if ((cv >> b) & 1) sv ^= w[b];
        tb.s[cv] = (uint8_t)sv;
    }
}

// rows / rbits: host arrays of device pointers; coef2 / mu2: canonical 2-limb host scalars
int ffgpu_launch_gf8_mask_open(const void* policy, int device, const void* const* rows, const uint64_t* coef2, int nrows,
                               const void* const* rbits, const uint64_t* mu2, int np, void* out, size_t n, hipStream_t st) {
    const GF2P8& f = *reinterpret_cast<const GF2P8*>(policy);
    if (nrows < 0 || nrows > GF8_MO_MAXROWS || np < 0 || np > GF8_MO_MAXP) return 2;
    static thread_local Gf8ByteTables tb;
    static thread_local uint32_t tb_for[3] = {~0u, 0, 0};
    if (tb_for[0] != f.n || tb_for[1] != f.red) {                  // sum_b 2^b x_b: depends on the field only
        gf8_byte_tables(f, nullptr, nullptr, tb);
        tb_for[0] = f.n;
        tb_for[1] = f.red;
    }
    Gf8MaskOpenArgs a;
    memset(&a, 0, sizeof(a));
    a.nrows = nrows;
    a.np = np;
    bool vec = (((uintptr_t)out) & 1u) == 0;
    for (int r = 0; r < nrows; ++r) {
        a.rows[r] = (const uint8_t*)rows[r];
        a.coef[r] = (uint32_t)(coef2[2 * r] & 0xffu);
        vec = vec && (((uintptr_t)rows[r]) & 1u) == 0;
    }
    for (int p = 0; p < np; ++p) {
        a.rbits[p] = (const uint8_t*)rbits[p];
        a.mu[p] = (uint32_t)(mu2[2 * p] & 0xffu);
        vec = vec && (((uintptr_t)rbits[p]) & 15u) == 0;
    }
    const size_t npair = vec ? n / 2 : 0;
    LaunchCfg lc = launch_cfg(device);
    unsigned grid = grid_for(npair ? (npair + 1) / 2 : n, lc);
    hipLaunchKernelGGL(k_gf8_mask_open, dim3(grid), dim3(BLOCK), 0, st, f, tb, a, (uint8_t*)out, npair, n);
    FFGPU_CHECK_LAUNCH();
    return 0;
}

int ffgpu_launch_gf8_bits_affine_fold(const void* policy, int device, const uint64_t* m2, const uint64_t* bias2, const void* c,
                                      const void* rbits, size_t ybr, void* out, size_t ybo, size_t n, int nbatch,
                                      hipStream_t st) {
    const GF2P8& f = *reinterpret_cast<const GF2P8*>(policy);
    Gf8ByteTables tb;
    gf8_byte_tables(f, m2, bias2, tb);
    bool vec = (((uintptr_t)c) & 1u) == 0 && (((uintptr_t)rbits) & 15u) == 0 && (((uintptr_t)out) & 1u) == 0;
    if (nbatch > 1) vec = vec && (ybr % 16 == 0) && (ybo % 2 == 0);
    const size_t npair = vec ? n / 2 : 0;
    LaunchCfg lc = launch_cfg(device);
    unsigned grid = grid_for(npair ? (npair + 1) / 2 : n, lc);
    hipLaunchKernelGGL(k_gf8_bits_affine_fold, dim3(grid, (unsigned)nbatch), dim3(BLOCK), 0, st, tb,
                       (const uint8_t*)c, (const uint8_t*)rbits, ybr, (uint8_t*)out, ybo, npair, n);
    FFGPU_CHECK_LAUNCH();
    return 0;
}

// ---- GF(2^8): the WHOLE secure S-box layer of all parties in one kernel (demos/np_aes.py:37-43) ------------------
// Every step of the layer is element-wise: byte h of the result depends on byte h of the parties' shares and on
// the 8 bit shares of position h only.  With all m parties of a computation on one GPU a thread can therefore carry
// the m shares of four bytes (one SWAR word per party) through the entire protocol in registers:
//   * x^254 by the reference's addition chain (runtime.py:1356-1367): 11 secure multiplications, each = local
//     products of the k = 2t+1 senders (log/antilog tables in LDS), their re-sharing with t fresh coefficients per
//     sender and byte from the in-register ChaCha stream, and the recipients' recombination (runtime.py:1096-1141,
//     603-689).  The recombination is linear, so party j's new share
//         sum_i lam_i (p_i + sum_q c_iq x_j^q)  =  (sum_i lam_i p_i) + sum_q (sum_i lam_i c_iq) x_j^q
//     is evaluated in the second form: the same field elements, K + K T constant products instead of K M;
//   * np_to_bits (runtime.py:4411-4423): c = open(y + r_modl) from the first t+1 parties, bits(c) + r_bits;
//   * the GF(2) affine map on the bit shares and np_from_bits (np_aes.py:40-42), both as byte-table look-ups.
// HBM traffic: the m share rows in, the m x 8 bit-share rows in, m rows out = 10 m bytes per secure byte instead of
// the 269 of the 13-kernel composition, and ONE launch -- the layer is bound by ChaCha and the GF(2^8) products.
struct Gf8SboxLayerArgs {
    const uint8_t* x;       // shares of the parties: row j at x + j * xs, n bytes
    const uint8_t* r;       // their shares of 8 random bits per byte: row j at r + j * rs, 8 n bytes
    uint8_t* out;           // row j at out + j * os
    size_t xs, rs, os;
    const uint8_t* tables;  // device: Gf8Tables (log / antilog), then Gf8ByteTables of np_from_bits, then of the affine fold
    uint32_t lam[7];        // Lagrange coefficients at 0 of the senders 1..2t+1
    uint32_t mu[4];         // ... of the opening parties 1..t+1
};
enum { SBL_TABLE_BYTES = 1536 + 2304 + 2304 };

// The keystream of the layer, two ways (same words in the same order: block b of word i has counter i * NBLK + b):
//   DR == 0  any round count: a block is computed when the previous one is used up -- a burst of ~1000 VALU
//            instructions between phases that wait on LDS look-ups;
//   DR  > 0  DR double rounds, known at compile time (10 = ChaCha20): block b + 1 advances by DR / 2 quarter rounds for
//            every word taken from block b, i.e. its arithmetic sits BETWEEN the table look-ups of the gates that
//            consume block b and fills their latency.  At 10^6 bytes every SIMD holds four waves that all start
//            together: without this they run their ChaCha bursts and their look-up phases in step (43 us against
//            24 us per 10^6 bytes in a long run, profiles/r04_sbox_layer.md).
template <int DR>
struct SblKeystream {
    uint32_t ks[16];        // block being consumed
    uint32_t x[16];         // DR > 0: the next block, in progress
    int kpos, qpos;
    uint64_t next_ctr;      // counter of the block in x (DR > 0) / of the next block to compute (DR == 0)
    const RngKey* rk;

    __device__ __forceinline__ void start_block(uint64_t ctr) {
        x[0] = 0x61707865u; x[1] = 0x3320646eu; x[2] = 0x79622d32u; x[3] = 0x6b206574u;
#pragma unroll
        for (int q = 0; q < 8; ++q) x[4 + q] = rk->key[q];
        x[12] = (uint32_t)ctr; x[13] = (uint32_t)(ctr >> 32); x[14] = rk->nonce[0]; x[15] = rk->nonce[1];
    }
    __device__ __forceinline__ void finish_block(uint64_t ctr) {       // ks = x + the block's input words
        ks[0] = x[0] + 0x61707865u; ks[1] = x[1] + 0x3320646eu; ks[2] = x[2] + 0x79622d32u; ks[3] = x[3] + 0x6b206574u;
#pragma unroll
        for (int q = 0; q < 8; ++q) ks[4 + q] = x[4 + q] + rk->key[q];
        ks[12] = x[12] + (uint32_t)ctr; ks[13] = x[13] + (uint32_t)(ctr >> 32);
        ks[14] = x[14] + rk->nonce[0]; ks[15] = x[15] + rk->nonce[1];
    }
    __device__ __forceinline__ void quarter(int q) {                    // q = index within a double round (constant-folded)
        switch (q & 7) {
            case 0: { FF_QR(x[0], x[4], x[8], x[12]) } break;
            case 1: { FF_QR(x[1], x[5], x[9], x[13]) } break;
            case 2: { FF_QR(x[2], x[6], x[10], x[14]) } break;
            case 3: { FF_QR(x[3], x[7], x[11], x[15]) } break;
            case 4: { FF_QR(x[0], x[5], x[10], x[15]) } break;
            case 5: { FF_QR(x[1], x[6], x[11], x[12]) } break;
            case 6: { FF_QR(x[2], x[7], x[8], x[13]) } break;
            default: { FF_QR(x[3], x[4], x[9], x[14]) } break;
        }
    }
    // begin(): x holds block `first_ctr` complete (DR > 0) / nothing is computed yet (DR == 0); step_begin() marks the
    // consumed block as used up.  All positions (kpos, qpos) are compile-time constants once the caller is unrolled:
    // a step that takes a multiple of 16 words from the stream leaves it in the state step_begin() describes.
    __device__ __forceinline__ void step_begin() {
        kpos = 16;
        qpos = 8 * DR;
    }
    __device__ __forceinline__ void begin(const RngKey* key, uint64_t first_ctr) {
        rk = key;
        next_ctr = first_ctr;
        if (DR > 0) {
            start_block(next_ctr);
#pragma unroll
            for (int q = 0; q < 8 * DR; ++q) quarter(q);
        }
        step_begin();
    }
    __device__ __forceinline__ uint32_t next_word() {
        if (DR == 0) {
            if (kpos == 16) {
                chacha_block(rk->key, (uint32_t)next_ctr, (uint32_t)(next_ctr >> 32), rk->nonce[0], rk->nonce[1], (int)rk->rounds, ks);
                ++next_ctr;
                kpos = 0;
            }
            return ks[kpos++];
        }
        if (kpos == 16) {                      // the block in x has had its 16 x DR / 2 = 8 DR quarter rounds: it becomes ks
            finish_block(next_ctr);
            ++next_ctr;
            start_block(next_ctr);
            kpos = 0;
            qpos = 0;
        }
        const uint32_t w = ks[kpos++];
#pragma unroll
        for (int q = 0; q < DR / 2; ++q) quarter(qpos++);
        return w;
    }
};

// W SWAR words (four secure bytes each) of one thread through the whole layer.  The words share ONE keystream: W x 33
// coefficient words (m = 3, t = 1) are 4.125 ChaCha blocks for W = 2 -- five blocks for eight bytes instead of six --
// and the 2 W K table products of a gate are independent of each other (twice the look-ups in flight per wave).
// full: every word of the thread holds four bytes of every row (aligned dword / 16-byte accesses); otherwise word w holds
// valid[w] in 0..4 bytes (the ragged end of the rows): byte accesses, nothing past a row's end.
template <int M, int T, int DR, int W, bool CONT>
__device__ __forceinline__ void sbl_words(const GF2P8& f, const Gf8SboxLayerArgs& a, const RngArgs& ra, const uint16_t* lg,
                                          const uint8_t* ex, const uint8_t* tbits, const uint8_t* tfold, size_t i, bool full,
                                          const int (&valid)[W], SblKeystream<DR>& stream, uint64_t step_ctr, uint32_t spare_word) {
    constexpr int K = 2 * T + 1;
    constexpr int NWORDS_RNG = 11 * K * T * W;                // keystream words per step
    static_assert(DR % 2 == 0, "DR / 2 quarter rounds per keystream word");
    static_assert(!CONT || NWORDS_RNG == 33, "continued keystream: two whole blocks + one spare word per step");
    // CONT: the thread's keystream continues from step to step -- 32 words = two whole blocks from the stream, the 33rd
    // from a spare block that serves 16 steps (spare_word).  Otherwise the step starts a stream of its own at step_ctr.
    if (CONT) stream.step_begin();
    else stream.begin(&ra.rk, step_ctr);
    int ndraw = 0;
    auto draw = [&]() -> uint32_t {
        if (CONT && ndraw == 32) {
            ++ndraw;
            return spare_word;
        }
        ++ndraw;
        return stream.next_word();
    };
    // constant product lam * w (lam wave-uniform): Horner over the bits of lam, scalar branches
    auto cmul = [&](uint32_t lam, uint32_t w) -> uint32_t {
        GF2P8::acc acc;
        f.acc_zero(acc);
        f.acc_mac(acc, lam, w);
        return f.acc_reduce(acc);
    };
    auto tabmul = [&](uint32_t u, uint32_t v) -> uint32_t {
        uint32_t acc = 0;
#pragma unroll
        for (int k8 = 0; k8 < 4; ++k8) {
            const uint32_t lsum = (uint32_t)lg[(u >> (8 * k8)) & 0xffu] + (uint32_t)lg[(v >> (8 * k8)) & 0xffu];
            acc |= (uint32_t)ex[lsum] << (8 * k8);
        }
        return acc;
    };
    uint32_t d[W][M], c[W][M], e[W][M];
    uint4 rb[W][M][2];
    if (full) {
#pragma unroll
        for (int w = 0; w < W; ++w)
#pragma unroll
            for (int j = 0; j < M; ++j) {
                d[w][j] = reinterpret_cast<const uint32_t*>(a.x + (size_t)j * a.xs)[i * W + w];
                const uint4* rv = reinterpret_cast<const uint4*>(a.r + (size_t)j * a.rs);
                rb[w][j][0] = ldg<true>(rv + 2 * (i * W + w));
                rb[w][j][1] = ldg<true>(rv + 2 * (i * W + w) + 1);
            }
    } else {
#pragma unroll
        for (int w = 0; w < W; ++w)
#pragma unroll
            for (int j = 0; j < M; ++j) {
                const uint8_t* xr = a.x + (size_t)j * a.xs + 4 * (i * W + w);
                const uint8_t* rr = a.r + (size_t)j * a.rs + 32 * (i * W + w);
                uint32_t xw = 0, rw[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
                for (int b = 0; b < 4; ++b) {                  // (fully unrolled: rw[] stays in registers)
                    if (b < valid[w]) {
                        xw |= (uint32_t)xr[b] << (8 * b);
#pragma unroll
                        for (int q = 0; q < 8; ++q) rw[2 * b + (q >> 2)] |= (uint32_t)rr[8 * b + q] << (8 * (q & 3));
                    }
                }
                d[w][j] = xw;
                rb[w][j][0] = make_uint4(rw[0], rw[1], rw[2], rw[3]);
                rb[w][j][1] = make_uint4(rw[4], rw[5], rw[6], rw[7]);
            }
    }
    // one secure multiplication for all parties and all W words: o <- shares of u * v
    auto gate = [&](const uint32_t (&u)[W][M], const uint32_t (&v)[W][M], uint32_t (&o)[W][M]) {
        uint32_t P[W], C[W][T];
#pragma unroll
        for (int w = 0; w < W; ++w) {
            P[w] = 0;
#pragma unroll
            for (int q = 0; q < T; ++q) C[w][q] = 0;
        }
#pragma unroll
        for (int s_ = 0; s_ < K; ++s_)
#pragma unroll
            for (int w = 0; w < W; ++w) {
                P[w] ^= cmul(a.lam[s_], tabmul(u[w][s_], v[w][s_]));
#pragma unroll
                for (int q = 0; q < T; ++q) C[w][q] ^= cmul(a.lam[s_], draw() & f.emask);
            }
        uint32_t res[W][M];
#pragma unroll
        for (int w = 0; w < W; ++w)
#pragma unroll
            for (int j = 0; j < M; ++j) {
                uint32_t h = C[w][T - 1];
#pragma unroll
                for (int q = T - 2; q >= 0; --q) h = f.muladd_small(h, (uint32_t)(j + 1), C[w][q]);
                res[w][j] = f.muladd_small(h, (uint32_t)(j + 1), P[w]);
            }
#pragma unroll
        for (int w = 0; w < W; ++w)
#pragma unroll
            for (int j = 0; j < M; ++j) o[w][j] = res[w][j];
    };
    auto copy = [&](uint32_t (&dst)[W][M], const uint32_t (&src)[W][M]) {
#pragma unroll
        for (int w = 0; w < W; ++w)
#pragma unroll
            for (int j = 0; j < M; ++j) dst[w][j] = src[w][j];
    };
    gate(d, d, c);          // x^2
    gate(c, c, c);          // x^4
    gate(c, c, c);          // x^8
    gate(c, d, c);          // x^9
    gate(c, c, c);          // x^18
    gate(c, d, e);          // x^19   (c, d = c*c, c*d: both from the old c)
    gate(c, c, c);          // x^36
    copy(d, e);
    gate(c, d, e);          // x^55
    gate(c, c, c);          // x^72
    copy(d, e);
    gate(c, d, c);          // x^127
    gate(c, c, c);          // x^254
#pragma unroll
    for (int w = 0; w < W; ++w) {
        // np_to_bits: open c + r_modl from the first t+1 parties
        uint32_t opened = 0;
#pragma unroll
        for (int p_ = 0; p_ <= T; ++p_) {
            const uint32_t rmod = gf8_tab_group(tbits, rb[w][p_][0].x, rb[w][p_][0].y) | (gf8_tab_group(tbits, rb[w][p_][0].z, rb[w][p_][0].w) << 8) |
                                  (gf8_tab_group(tbits, rb[w][p_][1].x, rb[w][p_][1].y) << 16) |
                                  (gf8_tab_group(tbits, rb[w][p_][1].z, rb[w][p_][1].w) << 24);
            opened ^= cmul(a.mu[p_], c[w][p_] ^ rmod);
        }
        // bits(opened) + r_bits -> affine map -> np_from_bits, per party
#pragma unroll
        for (int j = 0; j < M; ++j) {
            const uint32_t b0 = tfold[2048 + (opened & 0xffu)] ^ gf8_tab_group(tfold, rb[w][j][0].x, rb[w][j][0].y);
            const uint32_t b1 = tfold[2048 + ((opened >> 8) & 0xffu)] ^ gf8_tab_group(tfold, rb[w][j][0].z, rb[w][j][0].w);
            const uint32_t b2 = tfold[2048 + ((opened >> 16) & 0xffu)] ^ gf8_tab_group(tfold, rb[w][j][1].x, rb[w][j][1].y);
            const uint32_t b3 = tfold[2048 + (opened >> 24)] ^ gf8_tab_group(tfold, rb[w][j][1].z, rb[w][j][1].w);
            const uint32_t word = b0 | (b1 << 8) | (b2 << 16) | (b3 << 24);
            uint8_t* orow = a.out + (size_t)j * a.os;
            if (full) {
                reinterpret_cast<uint32_t*>(orow)[i * W + w] = word;
            } else {
#pragma unroll
                for (int b = 0; b < 4; ++b)
                    if (b < valid[w]) orow[4 * (i * W + w) + b] = (uint8_t)(word >> (8 * b));
            }
        }
    }
}

// nthreads_full threads own W whole words each; one more thread (the first of the next block slot) owns what is left:
// up to W - 1 whole words and the n % 4 bytes after them
template <int M, int T, int W>
__global__ __launch_bounds__(BLOCK) void k_gf8_sbox_layer(GF2P8 f, Gf8SboxLayerArgs a, RngArgs ra, size_t nthreads_full,
                                                          int rest_bytes) {
    __shared__ uint32_t lds32[SBL_TABLE_BYTES / 4];
    __shared__ uint32_t spare_lds[(T == 1 && W == 1) ? 16 * BLOCK : 1];      // the continued keystream's spare blocks
    for (int i = threadIdx.x; i < SBL_TABLE_BYTES / 4; i += BLOCK) lds32[i] = reinterpret_cast<const uint32_t*>(a.tables)[i];
    __syncthreads();
    const uint16_t* lg = reinterpret_cast<const uint16_t*>(lds32);
    const uint8_t* ex = reinterpret_cast<const uint8_t*>(lds32) + 512;
    const uint8_t* tbits = reinterpret_cast<const uint8_t*>(lds32) + 1536;
    const uint8_t* tfold = tbits + 2304;
    rng_load_state(ra);
    const size_t gid = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    const size_t gsz = (size_t)gridDim.x * BLOCK;
    int all4[W];
#pragma unroll
    for (int w = 0; w < W; ++w) all4[w] = 4;
    // ONE keystream per thread, continued from step to step (t = 1, one word per thread): 11 K T = 33 coefficient words per
    // step do not fill whole 16-word blocks, and a thread that starts a fresh block for every step discards 15 of every
    // 48 words -- 31 % of the layer's ChaCha work, which is ~90 % of its VALU instructions.  Here a step takes two whole
    // blocks from the thread's stream (compile-time positions) and its 33rd word from a spare block that serves 16
    // steps; the launcher caps the grid at the resident workgroups, so at 10^8 bytes a thread walks through ~95 steps
    // on 2.06 blocks each.  Counters: thread g streams blocks [g * bpt, (g + 1) * bpt), its spare blocks are
    // 2^62 + g * spt + (step / 16), the ragged end of the rows draws from 2^61.
    constexpr bool CAN_CONT = (T == 1 && W == 1);
    constexpr int NBLK = (11 * (2 * T + 1) * T * W + 15) / 16;
    const uint64_t steps = (nthreads_full + gsz - 1) / gsz;
    const uint64_t bpt = 2 * steps + 2, spt = steps / 16 + 1;
    auto run = [&](auto dr_, auto cont_) {
        constexpr int DRV = decltype(dr_)::value;
        constexpr bool CONT = decltype(cont_)::value;
        SblKeystream<DRV> stream;
        if (CONT && gid < nthreads_full) stream.begin(&ra.rk, (uint64_t)gid * bpt);
        uint64_t j = 0;
        for (size_t i = gid; i < nthreads_full; i += gsz, ++j) {
            uint32_t sw = 0;
            if constexpr (CONT) {
                // the spare block lives in LDS ([word][thread]: conflict-free), not in 16 registers: the kernel stays at
                // four waves per SIMD
                if ((j & 15) == 0) {
                    const uint64_t c = (1ull << 62) + (uint64_t)gid * spt + (j >> 4);
                    uint32_t blk[16];
                    chacha_block(ra.rk.key, (uint32_t)c, (uint32_t)(c >> 32), ra.rk.nonce[0], ra.rk.nonce[1], (int)ra.rk.rounds, blk);
#pragma unroll
                    for (int q = 0; q < 16; ++q) spare_lds[q * BLOCK + threadIdx.x] = blk[q];
                }
                sw = spare_lds[(int)(j & 15) * BLOCK + threadIdx.x];
            }
            sbl_words<M, T, DRV, W, CONT>(f, a, ra, lg, ex, tbits, tfold, i, true, all4, stream, (uint64_t)i * NBLK, sw);
        }
    };
    // (a thread with one or two steps gains nothing from the continued stream and would compute its spare block as a
    // burst: 45.6 us against 38.5 us at 10^6 bytes, measured -- those launches take a fresh stream per step)
#if defined(SBL_PROBE_PATH)      // static instruction counts of ONE path (tools: hipcc -DSBL_PROBE_PATH=n -save-temps)
    const bool cont = CAN_CONT && (SBL_PROBE_PATH & 1);
    const bool dr10 = (SBL_PROBE_PATH & 2) != 0;
#else
    const bool cont = CAN_CONT && steps >= 4;
    const bool dr10 = ra.rk.rounds == 20;
#endif
    if constexpr (CAN_CONT) {
        if (cont) {
            if (dr10) run(std::integral_constant<int, 10>(), std::true_type());
            else run(std::integral_constant<int, 0>(), std::true_type());
        }
    }
    if (!cont) {
        if (dr10) run(std::integral_constant<int, 10>(), std::false_type());
        else run(std::integral_constant<int, 0>(), std::false_type());
    }
    if (rest_bytes && gid == 0) {                                // the ragged end of every row
        int valid[W];
#pragma unroll
        for (int w = 0; w < W; ++w) {
            const int left = rest_bytes - 4 * w;
            valid[w] = left >= 4 ? 4 : (left > 0 ? left : 0);
        }
        SblKeystream<0> stream;
        sbl_words<M, T, 0, W, false>(f, a, ra, lg, ex, tbits, tfold, nthreads_full, false, valid, stream, 1ull << 61, 0u);
    }
    rng_state_release(ra);
}

// tables_dev: SBL_TABLE_BYTES of device memory (the caller caches it per matrix); returns 2 when the shape is not
// covered (rows not 4- / 16-byte aligned, m > 7 or t > 3): the caller composes the layer from the per-step kernels
// then.  Any n: the n % 4 bytes after the last whole word of each row are handled by one thread with byte accesses.
int ffgpu_launch_gf8_sbox_layer(const void* policy, int device, const void* x, size_t xs, const void* r, size_t rs, void* out,
                                size_t os, const void* tables_dev, const uint64_t* lam2, const uint64_t* mu2, int t, int m,
                                size_t n, hipStream_t st, const RngArgs* rng) {
    const GF2P8& f = *reinterpret_cast<const GF2P8*>(policy);
    if (f.n != 8 || (((uintptr_t)x | (uintptr_t)out | xs | os) & 3) || (((uintptr_t)r | rs) & 15)) return 2;
    Gf8SboxLayerArgs a;
    memset(&a, 0, sizeof(a));
    a.x = (const uint8_t*)x; a.r = (const uint8_t*)r; a.out = (uint8_t*)out;
    a.xs = xs; a.rs = rs; a.os = os;
    a.tables = (const uint8_t*)tables_dev;
    for (int i = 0; i < 2 * t + 1; ++i) a.lam[i] = (uint32_t)(lam2[2 * i] & 0xffu);
    for (int i = 0; i <= t; ++i) a.mu[i] = (uint32_t)(mu2[2 * i] & 0xffu);
    RngArgs ra = *rng;
    // One word (four bytes) per thread.  (Two words per thread sharing a keystream -- five ChaCha blocks for eight bytes instead
    // of six, but 131 VGPRs and half the waves -- measured slower at 10^6 bytes in round 4 and was removed; the continued
    // keystream of the kernel gets the same saving without the registers: profiles/r04_sbox_layer.md.)
    constexpr int W = 1;
    const size_t nthreads_full = n / (4 * (size_t)W);
    const int rest_bytes = (int)(n - nthreads_full * 4 * W);
    size_t want = (nthreads_full + BLOCK - 1) / BLOCK;
    if (want < 1) want = 1;
    // persistent above one round of resident workgroups (4 per CU at t = 1, m <= 4: 119-128 VGPRs; 3 for m >= 5; 2-3 per CU
    // for t >= 2): the threads of the t = 1 kernels then carry their keystream from step to step instead of discarding 15
    // of every 48 words
    LaunchCfg lc = launch_cfg(device);
    const size_t resident = (size_t)(lc.num_cu > 0 ? lc.num_cu : 256) * (t == 1 ? (m <= 4 ? 4 : 3) : (t == 2 ? 3 : 2));
    if (want > resident) want = resident;
    const unsigned grid = (unsigned)want;
    {   // the keystream's counter ranges never meet: threads stream blocks [g * bpt, (g + 1) * bpt) (or step i its blocks
        // i * NBLK ...), the ragged end draws from 2^61, the spare blocks of the continued stream from 2^62 + g * spt + j / 16,
        // re-draws elsewhere in the library from 2^63 -- refuse shapes that would leave their range (n > ~10^17 bytes)
        const uint64_t gsz = (uint64_t)grid * BLOCK;
        const uint64_t steps = (nthreads_full + gsz - 1) / gsz;
        if (gsz * (2 * steps + 2) >= (1ull << 61) || gsz * (steps / 16 + 1) >= (1ull << 61) || nthreads_full * 16ull >= (1ull << 61)) return 2;
    }
    // (measured and NOT kept, round 4: letting up to 1024 workgroups advance the state themselves with a ticket drawn at
    // workgroup START -- the returning atomic sits in front of the wave's first loads in the in-order vmcnt queue, and 977
    // same-address atomics at ~25 ns each delay those waves: 43.7 us against 38.4 us with the one-thread kernel after it)
    ra.release = (ra.dev_key && !ra.no_advance && grid <= (unsigned)RNG_RELEASE_MAX_GRID) ? 1 : 0;
#define SBL_LAUNCH(MM, TT, WW)                                                                                        \
    {                                                                                                                 \
        hipLaunchKernelGGL((k_gf8_sbox_layer<MM, TT, WW>), dim3(grid), dim3(BLOCK), 0, st, f, a, ra, nthreads_full,   \
                           rest_bytes);                                                                               \
        FFGPU_CHECK_LAUNCH();                                                                                         \
        if (ra.dev_key && !ra.release && !ra.no_advance)                                                              \
            hipLaunchKernelGGL((k_rng_advance<0>), dim3(1), dim3(1), 0, st, const_cast<RngKey*>(ra.dev_key), 1u);     \
        return 0;                                                                                                     \
    }
#define SBL_CASE(MM, TT)                                 \
    if (m == MM && t == TT) {                            \
        SBL_LAUNCH(MM, TT, 1)                            \
    }
    SBL_CASE(3, 1) SBL_CASE(4, 1) SBL_CASE(5, 1) SBL_CASE(5, 2) SBL_CASE(6, 1) SBL_CASE(6, 2) SBL_CASE(7, 1) SBL_CASE(7, 2)
    SBL_CASE(7, 3)
#undef SBL_CASE
#undef SBL_LAUNCH
    return 2;
}

// host: the three tables of the layer in the layout the kernel expects
void ffgpu_gf8_sbox_layer_tables(const void* policy, const void* mul_tables, const uint64_t* m2, const uint64_t* bias2,
                                 unsigned char* out) {
    const GF2P8& f = *reinterpret_cast<const GF2P8*>(policy);
    memcpy(out, mul_tables, 1536);
    Gf8ByteTables tb;
    gf8_byte_tables(f, nullptr, nullptr, tb);
    memcpy(out + 1536, &tb, 2304);
    gf8_byte_tables(f, m2, bias2, tb);
    memcpy(out + 1536 + 2304, &tb, 2304);
}

// ---- GF(2^64), x^64 + x^4 + x^3 + x + 1: element-wise product, BIT-SLICED ------------------------------------------
// (gfpx.py:988-1045 + finfields.py:537-541.)  The multiplier route (ff_clmul64: 48 v_mad_u64_u32 + masks, 280 VALU
// instructions per element) is issue-bound at 0.36 of the HBM rate.  Bit-sliced, a partial-product MAC of a whole batch of
// elements is ONE v_bitop3_b32 (acc ^ (a & b)) and Karatsuba runs down to 8 x 8 leaves.  Round 4 held 32 elements per lane as
// 64 full-width planes per operand: 455 registers, ONE wave per SIMD, VALU busy 0.65 (nothing covers the dependent-issue
// latency), 66-70 us at n = 10^7.  Round 5 (bitslice.hpp, mul16_planes): 16 elements per lane with the two 32-coefficient
// HALVES of an operand packed into the two halves of a plane register -- one 32 x 32 transpose of [lo words ; hi words] yields
// exactly that -- so the outer Karatsuba products A0 B0 and A1 B1 come out of ONE Mul<32> on packed registers, the middle
// product is packed the same way one level down, and nothing is ever held as 64 or 127 full-width planes: the kernel fits
// TWO waves per SIMD (amdgpu_waves_per_eu(2, 2)): 214 VGPRs, 58.7 us at n = 10^7 (0.51 of the HBM rate; round 4: 66-70 us).
// One slab per wave on an uncapped grid: a finished wave is replaced at once and the new wave's loads overlap its partner's
// arithmetic.  (Measured and NOT kept: a persistent variant that streams the next slab's operands global -> LDS with
// global_load_lds while the current one is multiplied -- 72.3 us: sixteen LDS-DMA pieces cost more issue time than the
// latency they hide, and the capped grid loses the tail.)  What still separates the kernel from its issue bound (~42 us for
// this instruction mix: three-operand instructions take ~4.1 cycles per wave64, two-operand ones 2.4 -- profiles/
// r05_valu_rates.md) is the time a wave runs ALONE on its SIMD while its partner waits for memory: a lone wave issues at most
// one instruction per 4.7 cycles.  A third wave needs <= 168 registers: forced with amdgpu_waves_per_eu(3, 3) the compiler
// spills 33 registers to scratch and the kernel takes 74.5 us -- measured, not kept.  The same arithmetic compiles with g++
// (tests/test_hostcheck.py).
// A slab = 1024 consecutive elements = 512 uint4; lane l of the wave that owns it reads uint4 number r * 64 + l
// (coalesced), i.e. holds elements 128 r + 2 l and 128 r + 2 l + 1, r = 0..7.  The last slab may be partial: its accesses
// are guarded per uint4 (nvec4 = n / 2 of them exist), missing operands are zero.  In place (o == a or o == b) is fine: a
// wave loads its whole slab before it stores any of it, and slabs are disjoint -- hence no __restrict__ on the pointers.
__global__ __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu(2, 2)))
void k_gf2w64_mul_bitsliced(const uint4* a, const uint4* b, uint4* o, size_t nslab, size_t nvec4) {
    const int lane = threadIdx.x & 63;
    const size_t slab = (size_t)blockIdx.x * (BLOCK / 64) + (threadIdx.x >> 6);
    if (slab >= nslab) return;
    const size_t base = slab * 512 + lane;
    const bool whole = (slab + 1) * 512 <= nvec4;
    uint32_t pa[32], pb[32];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const bool in = whole || base + (size_t)r * 64 < nvec4;
        const uint4 va = in ? ldg<true>(a + base + (size_t)r * 64) : make_uint4(0, 0, 0, 0);
        const uint4 vb = in ? ldg<true>(b + base + (size_t)r * 64) : make_uint4(0, 0, 0, 0);
        pa[2 * r] = va.x; pa[16 + 2 * r] = va.y; pa[2 * r + 1] = va.z; pa[16 + 2 * r + 1] = va.w;      // rows: lo words, then hi words
        pb[2 * r] = vb.x; pb[16 + 2 * r] = vb.y; pb[2 * r + 1] = vb.z; pb[16 + 2 * r + 1] = vb.w;
    }
    bs64::transpose32(pa);
    bs64::transpose32(pb);
    uint32_t c[127];
    bs64::mul16_planes(pa, pb, c);
    bs64::fold_1b(c);                                      // x^64 = x^4 + x^3 + x + 1
    uint32_t ov[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) ov[i] = bs64::lo_pair(c[i + 32], c[i]);
    bs64::transpose32(ov);
#pragma unroll
    for (int r = 0; r < 8; ++r)
        if (whole || base + (size_t)r * 64 < nvec4)
            stg<true>(o + base + (size_t)r * 64, make_uint4(ov[2 * r], ov[16 + 2 * r], ov[2 * r + 1], ov[16 + 2 * r + 1]));
}

// returns the number of leading elements it has multiplied (all of them but the last one of an odd n; 0 = not applicable):
// the caller sends the rest through the element-wise kernel
size_t ffgpu_launch_gf2w64_mul_bitsliced(const void* policy, int device, const void* a, const void* b, void* out, size_t n,
                                         hipStream_t st) {
    const GF2W64& f = *reinterpret_cast<const GF2W64*>(policy);
    if (f.n != 64 || f.red != 0x1bull || n < ((size_t)1 << 21) || (((uintptr_t)a | (uintptr_t)b | (uintptr_t)out) & 15)) return 0;
    const size_t nvec4 = n / 2;
    const size_t nslab = (nvec4 + 511) / 512;
    const unsigned grid = (unsigned)((nslab + BLOCK / 64 - 1) / (BLOCK / 64));
    hipLaunchKernelGGL(k_gf2w64_mul_bitsliced, dim3(grid), dim3(BLOCK), 0, st, (const uint4*)a, (const uint4*)b, (uint4*)out, nslab,
                       nvec4);
    return nvec4 * 2;
}

// ---- GF(2^n), n <= 8: multiplication through log / antilog tables in LDS ---------------------
// c = exp[log a + log b].  log[0] = 2*(q-1) so that any sum involving a zero operand lands in the
// zero-padded tail of exp[] (no zero test, no select).  Both tables come from the host (built at
// context creation with the same packed shift-xor arithmetic) as a kernel argument; each workgroup
// copies them to LDS once (1.5 KiB) and then streams 16 bytes per lane.
// log: 256 x u16 (512 B), exp: 4*(q-1)+1 <= 1021 x u8 -> 1.5 KiB of LDS, (nearly) conflict-free.
struct Gf8Tables {
    uint16_t lg[256];
    uint8_t ex[1024];
};

__global__ __launch_bounds__(BLOCK) void k_gf8_mul_tab(Gf8Tables tb, const uint8_t* __restrict__ a,
                                                        const uint8_t* __restrict__ b, uint8_t* __restrict__ out,
                                                        size_t nvec, size_t n) {
    __shared__ uint16_t lg[256];
    __shared__ uint8_t ex[1024];
    {
        uint32_t t = threadIdx.x;
        lg[t] = tb.lg[t];
        reinterpret_cast<uint32_t*>(ex)[t] = reinterpret_cast<const uint32_t*>(tb.ex)[t];
    }
    __syncthreads();
    const size_t gid = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    const size_t gsz = (size_t)gridDim.x * BLOCK;
    const uint4* __restrict__ av = reinterpret_cast<const uint4*>(a);
    const uint4* __restrict__ bv = reinterpret_cast<const uint4*>(b);
    uint4* __restrict__ ov = reinterpret_cast<uint4*>(out);
    for (size_t i = gid; i < nvec; i += gsz) {
        uint4 x = ldg<true>(av + i);
        uint4 y = ldg<true>(bv + i);
        uint32_t xw[4] = {x.x, x.y, x.z, x.w}, yw[4] = {y.x, y.y, y.z, y.w}, r[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            uint32_t acc = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                uint32_t s = (uint32_t)lg[(xw[q] >> (8 * k)) & 0xff] + (uint32_t)lg[(yw[q] >> (8 * k)) & 0xff];
                acc |= (uint32_t)ex[s] << (8 * k);
            }
            r[q] = acc;
        }
        stg<true>(ov + i, make_uint4(r[0], r[1], r[2], r[3]));
    }
    for (size_t e = nvec * 16 + gid; e < n; e += gsz) out[e] = ex[(uint32_t)lg[a[e]] + (uint32_t)lg[b[e]]];
}

// host: build the tables for GF(2^n) (any irreducible modulus of degree n <= 8)
int ffgpu_gf8_build_tables(const void* policy, void* tables_out) {
    const GF2P8& f = *reinterpret_cast<const GF2P8*>(policy);
    Gf8Tables* tb = reinterpret_cast<Gf8Tables*>(tables_out);
    const uint32_t q1 = (1u << f.n) - 1;  // multiplicative order
    memset(tb, 0, sizeof(*tb));
    if (q1 == 1) {  // GF(2): 1*1 = 1
        tb->lg[0] = 2;
        tb->lg[1] = 0;
        tb->ex[0] = 1;
        return 0;
    }
    for (uint32_t g = 2; g <= q1; ++g) {
        // order of g
        uint32_t x = 1, ord = 0;
        do {
            x = f.mul(x, g) & 0xffu;
            ++ord;
        } while (x != 1 && ord <= q1);
        if (ord != q1) continue;
        x = 1;
        for (uint32_t i = 0; i < q1; ++i) {
            tb->ex[i] = (uint8_t)x;
            tb->ex[i + q1] = (uint8_t)x;  // sums of two logs reach 2*(q1-1)
            tb->lg[x] = (uint16_t)i;
            x = f.mul(x, g) & 0xffu;
        }
        tb->lg[0] = (uint16_t)(2 * q1);   // 2*q1 .. 4*q1 stay zero in ex[]
        for (uint32_t v = q1 + 1; v < 256; ++v) tb->lg[v] = (uint16_t)(2 * q1);  // non-canonical bytes -> 0
        return 0;
    }
    return 1;
}

int ffgpu_launch_gf8_mul_tab(const void* tables, int device, const void* a, const void* b, void* out, size_t n,
                             hipStream_t st) {
    const Gf8Tables& tb = *reinterpret_cast<const Gf8Tables*>(tables);
    LaunchCfg lc = launch_cfg(device);
    bool vec = aligned16(a) && aligned16(b) && aligned16(out);
    size_t nvec = vec ? n / 16 : 0;
    unsigned grid = grid_for(nvec ? nvec : n, lc);      // uncapped: the per-workgroup table copy is cheap (measured best)
    hipLaunchKernelGGL(k_gf8_mul_tab, dim3(grid), dim3(BLOCK), 0, st, tb, (const uint8_t*)a, (const uint8_t*)b,
                       (uint8_t*)out, nvec, n);
    FFGPU_CHECK_LAUNCH();
    return 0;
}

// ---- GF(2^n), 9 <= n <= 128: 4-bit window multiplication with tables in LDS --------------------
// Per element: T[u] = a*u mod f for the 16 four-bit polynomials u (3 doublings + 11 xors), kept in
// LDS in a column-per-thread layout T[u][tid] (the bank of an entry does not depend on u, so the
// data-dependent look-ups are conflict-free); then b is consumed a nibble at a time from the top:
//     acc = acc * x^4 mod f          (shift by 4, overflow nibble o folded back through R[o])
//     acc ^= T[nibble]
// R[o] = o * x^n mod f (16 entries, host-built, 256 B in LDS: one entry per 4 banks, conflict-free).
// ~20 VALU ops + 2 LDS reads per nibble instead of ~25 ops per BIT for the shift-xor loop.
struct Gf2wRTable {
    uint64_t lo[16], hi[16];
};

template <int LIMBS>   // 1: uint64 elements (n <= 64), 2: u128e elements (n <= 128)
struct Gf2wTraits;
template <>
struct Gf2wTraits<2> {
    typedef GF2W128 F;
    typedef u128e E;
    typedef uint4 L;   // LDS entry
    static __device__ __forceinline__ L pack(uint64_t lo, uint64_t hi) {
        return make_uint4((uint32_t)lo, (uint32_t)(lo >> 32), (uint32_t)hi, (uint32_t)(hi >> 32));
    }
    static __device__ __forceinline__ void unpack(const L& v, uint64_t& lo, uint64_t& hi) {
        lo = (uint64_t)v.x | ((uint64_t)v.y << 32);
        hi = (uint64_t)v.z | ((uint64_t)v.w << 32);
    }
};
template <>
struct Gf2wTraits<1> {
    typedef GF2W64 F;
    typedef uint64_t E;
    typedef uint2 L;
    static __device__ __forceinline__ L pack(uint64_t lo, uint64_t) { return make_uint2((uint32_t)lo, (uint32_t)(lo >> 32)); }
    static __device__ __forceinline__ void unpack(const L& v, uint64_t& lo, uint64_t& hi) {
        lo = (uint64_t)v.x | ((uint64_t)v.y << 32);
        hi = 0;
    }
};

template <int LIMBS, bool SPARSE>   // SPARSE: modulus = x^n + r(x) with r < 2^28: fold the overflow nibble in registers
__global__ __launch_bounds__(BLOCK) void k_gf2w_mul_win(typename Gf2wTraits<LIMBS>::F f, Gf2wRTable rt,
                                                         const typename Gf2wTraits<LIMBS>::E* __restrict__ a,
                                                         const typename Gf2wTraits<LIMBS>::E* __restrict__ b,
                                                         typename Gf2wTraits<LIMBS>::E* __restrict__ out, size_t n) {
    typedef Gf2wTraits<LIMBS> Tr;
    typedef typename Tr::L L;
    __shared__ L T[16][BLOCK];
    __shared__ L R[16];
    const uint32_t tid = threadIdx.x;
    if (tid < 16) R[tid] = Tr::pack(rt.lo[tid], rt.hi[tid]);
    __syncthreads();
    const uint32_t deg = f.n;
    const size_t gid = (size_t)blockIdx.x * BLOCK + tid;
    const size_t gsz = (size_t)gridDim.x * BLOCK;
    for (size_t i = gid; i < n; i += gsz) {
        uint64_t alo, ahi, blo, bhi;
        if constexpr (LIMBS == 2) {
            u128e av = a[i], bv = b[i];
            alo = av.lo; ahi = av.hi; blo = bv.lo; bhi = bv.hi;
        } else {
            alo = a[i]; ahi = 0; blo = b[i]; bhi = 0;
        }
        // multiples of a: x1, x2, x4, x8 by doubling, the rest by xor
        uint64_t l1 = alo, h1 = ahi, l2, h2, l4, h4, l8, h8;
        auto dbl = [&](uint64_t lo, uint64_t hi, uint64_t& olo, uint64_t& ohi) {
            if constexpr (LIMBS == 2) {
                olo = lo; ohi = hi;
                f.xtime(olo, ohi);
            } else {
                olo = f.xtime(lo); ohi = 0;
            }
        };
        dbl(l1, h1, l2, h2);
        dbl(l2, h2, l4, h4);
        dbl(l4, h4, l8, h8);
        T[0][tid] = Tr::pack(0, 0);
        T[1][tid] = Tr::pack(l1, h1);
        T[2][tid] = Tr::pack(l2, h2);
        T[3][tid] = Tr::pack(l2 ^ l1, h2 ^ h1);
        T[4][tid] = Tr::pack(l4, h4);
        T[5][tid] = Tr::pack(l4 ^ l1, h4 ^ h1);
        T[6][tid] = Tr::pack(l4 ^ l2, h4 ^ h2);
        T[7][tid] = Tr::pack(l4 ^ l2 ^ l1, h4 ^ h2 ^ h1);
        T[8][tid] = Tr::pack(l8, h8);
        T[9][tid] = Tr::pack(l8 ^ l1, h8 ^ h1);
        T[10][tid] = Tr::pack(l8 ^ l2, h8 ^ h2);
        T[11][tid] = Tr::pack(l8 ^ l2 ^ l1, h8 ^ h2 ^ h1);
        T[12][tid] = Tr::pack(l8 ^ l4, h8 ^ h4);
        T[13][tid] = Tr::pack(l8 ^ l4 ^ l1, h8 ^ h4 ^ h1);
        T[14][tid] = Tr::pack(l8 ^ l4 ^ l2, h8 ^ h4 ^ h2);
        T[15][tid] = Tr::pack(l8 ^ l4 ^ l2 ^ l1, h8 ^ h4 ^ h2 ^ h1);
        uint64_t clo = 0, chi = 0;
        // All nibbles of the limb width are processed (leading zero nibbles of b only shift a zero
        // accumulator), so the trip count is a compile-time constant: the loop is fully unrolled and the
        // T look-ups, which do not depend on the accumulator, are issued 8 at a time ahead of their use.
        constexpr int NIB = LIMBS * 16;
#pragma unroll
        for (int kb = NIB - 8; kb >= 0; kb -= 8) {
            L tv[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int k = kb + 7 - j;
                uint32_t nib = (k >= 16) ? (uint32_t)(bhi >> (4 * (k - 16))) & 15u : (uint32_t)(blo >> (4 * k)) & 15u;
                tv[j] = T[nib][tid];
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                // acc *= x^4: overflow nibble = bits deg-4 .. deg-1 before the shift
                uint32_t o;
                if constexpr (LIMBS == 2) {
                    const uint32_t sh = deg - 4;            // 61 .. 124
                    o = sh >= 64 ? (uint32_t)(chi >> (sh - 64)) & 15u
                                 : (uint32_t)(((clo >> sh) | (chi << (64 - sh))) & 15u);
                    chi = ((chi << 4) | (clo >> 60)) & f.emask_hi;
                    clo <<= 4;
                } else {
                    o = (uint32_t)(clo >> (deg - 4)) & 15u;
                    clo = (clo << 4) & f.emask;
                }
                uint64_t rlo, rhi, tlo, thi;
                if constexpr (SPARSE) {
                    // o * r(x), carry-less, r < 2^28: stays in the low 32 bits (off the LDS latency path)
                    const uint32_t r0 = (uint32_t)rt.lo[1];
                    uint32_t rr = ((0u - (o & 1u)) & r0) ^ ((0u - ((o >> 1) & 1u)) & (r0 << 1)) ^
                                  ((0u - ((o >> 2) & 1u)) & (r0 << 2)) ^ ((0u - ((o >> 3) & 1u)) & (r0 << 3));
                    rlo = rr;
                    rhi = 0;
                } else {
                    Tr::unpack(R[o], rlo, rhi);
                }
                Tr::unpack(tv[j], tlo, thi);
                clo ^= rlo ^ tlo;
                chi ^= rhi ^ thi;
            }
        }
        if constexpr (LIMBS == 2) {
            u128e r;
            r.lo = clo;
            r.hi = chi;
            out[i] = r;
        } else {
            out[i] = clo;
        }
    }
}

// host: R[o] = o * x^n mod f for the 16 nibbles o
int ffgpu_gf2w_build_rtable(const void* policy, int limbs, void* rtable_out) {
    Gf2wRTable* rt = reinterpret_cast<Gf2wRTable*>(rtable_out);
    if (limbs == 2) {
        const GF2W128& f = *reinterpret_cast<const GF2W128*>(policy);
        for (uint32_t o = 0; o < 16; ++o) {
            uint64_t lo = o, hi = 0;
            for (uint32_t k = 0; k < f.n; ++k) f.xtime(lo, hi);
            rt->lo[o] = lo;
            rt->hi[o] = hi;
        }
    } else {
        const GF2W64& f = *reinterpret_cast<const GF2W64*>(policy);
        for (uint32_t o = 0; o < 16; ++o) {
            uint64_t v = o;
            if (f.n < 4) v = f.reduce_raw(v);
            for (uint32_t k = 0; k < f.n; ++k) v = f.xtime(v);
            rt->lo[o] = v;
            rt->hi[o] = 0;
        }
    }
    return 0;
}

int ffgpu_launch_gf2w_mul_win(const void* policy, int limbs, const void* rtable, int device, const void* a,
                              const void* b, void* out, size_t n, hipStream_t st) {
    const Gf2wRTable& rt = *reinterpret_cast<const Gf2wRTable*>(rtable);
    LaunchCfg lc = launch_cfg(device);
    unsigned grid = grid_for(n, lc);
    // R[1] = x^n mod f = r(x): sparse path when it fits 28 bits (every default MPyC irreducible does)
    const bool sparse = rt.hi[1] == 0 && rt.lo[1] < (1ull << 28);
    if (limbs == 2) {
        const GF2W128& f = *reinterpret_cast<const GF2W128*>(policy);
        if (sparse)
            hipLaunchKernelGGL((k_gf2w_mul_win<2, true>), dim3(grid), dim3(BLOCK), 0, st, f, rt, (const u128e*)a,
                               (const u128e*)b, (u128e*)out, n);
        else
            hipLaunchKernelGGL((k_gf2w_mul_win<2, false>), dim3(grid), dim3(BLOCK), 0, st, f, rt, (const u128e*)a,
                               (const u128e*)b, (u128e*)out, n);
    } else {
        const GF2W64& f = *reinterpret_cast<const GF2W64*>(policy);
        if (sparse)
            hipLaunchKernelGGL((k_gf2w_mul_win<1, true>), dim3(grid), dim3(BLOCK), 0, st, f, rt, (const uint64_t*)a,
                               (const uint64_t*)b, (uint64_t*)out, n);
        else
            hipLaunchKernelGGL((k_gf2w_mul_win<1, false>), dim3(grid), dim3(BLOCK), 0, st, f, rt, (const uint64_t*)a,
                               (const uint64_t*)b, (uint64_t*)out, n);
    }
    FFGPU_CHECK_LAUNCH();
    return 0;
}

// LDS read at an ABSOLUTE byte offset of the workgroup's allocation (ds_read with the constant part of the address
// as the instruction's immediate; a pointer derived from the `extern __shared__` symbol costs one v_add per access)
template <class L>
__device__ __forceinline__ L lds_entry(uint32_t byte_offset) {
    typedef uint32_t vec __attribute__((ext_vector_type(sizeof(L) / 4)));
    const vec v = *reinterpret_cast<const __attribute__((address_space(3))) vec*>(byte_offset);
    L r;
    __builtin_memcpy(&r, &v, sizeof(L));
    return r;
}
// byte offset of an LDS object inside the workgroup's allocation (0 for the only / first one: folds to a constant)
__device__ __forceinline__ uint32_t lds_offset_of(const void* p) {
    return (uint32_t)(size_t)(const __attribute__((address_space(3))) char*)(const char*)p;
}

// ---- GF(2^n), 9 <= n <= 128: recombination through shared nibble tables -----------------------------
// out[h] = sum_j lambda_j * rows[j][h].  The Lagrange coefficients are wave-uniform, and x -> lambda*x
// is GF(2)-linear, so each workgroup first builds, for every row j with a DENSE coefficient, nibble position pos
// and nibble value v, the entry  T[j][pos][v] = lambda_j * (v * x^(4 pos)) mod f  (KT * NPOS * 16 entries, built
// cooperatively with the in-register carry-less product), and then every element costs NPOS look-ups
// + xors per row instead of a full field multiplication.  An entry's bank depends only on v (16
// consecutive 16-byte slots per position), so the data-dependent reads are conflict-free
// (SQ_LDS_BANK_CONFLICT = 0, profiles/r03_gf2w.md).
//
// Rows whose coefficient is 1 are XORed in without any table (and rows with coefficient 0 are dropped by the
// launcher): when the interpolation points together with 0 are closed under XOR -- parties 1..3 or 1..7, i.e. every
// recombination of 2t+1 = m rows in `_reshare` for m = 3, 7 (runtime.py:658-661) -- ALL Lagrange coefficients at 0
// are 1 (prod_{l != j} x_l / (x_l + x_j) runs over the same set in numerator and denominator), and the
// recombination is a plain XOR of the rows at the HBM rate.
//
// Round 3: the look-ups of a row are issued 16 at a time into registers before any of them is consumed (the
// round-2 kernel waited after every second ds_read: 53 % of its wave cycles were s_waitcnt, SQ_WAIT_ANY), the
// nibble -> byte-offset conversion is one masked byte select per look-up (pre-shifted copies of the words), results
// are folded with 3-input XORs, and 512-thread workgroups share one table (16 waves per CU at k = 7).
enum { REC_BLOCK = 512, REC_MAXK = 9 };

template <int LIMBS>
struct Gf2wRecArgs {
    const void* trow[REC_MAXK];      // one row per GROUP of rows that share a dense coefficient ...
    const void* erow[REC_MAXK][3];   // ... and up to three more rows of the group: XORed first, they go through ONE table
    int gcnt[REC_MAXK];              // rows in group g (1..4).  Over GF(2^n) the coefficients of parties 1..5 at 0 are two
    //                                  distinct dense values twice each and a 1.  (All indices into these arrays are
    //                                  compile-time constants: a runtime index would demote the arguments to scratch.)
    const void* prow[REC_MAXK];      // rows with coefficient 1 (plain XOR)
    uint64_t lam_lo[REC_MAXK], lam_hi[REC_MAXK];     // per group
    int kp;
};

template <int LIMBS, int KT, bool DEEP>      // DEEP: register budget of 4 waves per SIMD -> whole batches of look-ups in flight
__global__ __launch_bounds__(REC_BLOCK) __attribute__((amdgpu_waves_per_eu(DEEP ? 4 : 1, DEEP ? 4 : 8))) void k_gf2w_recombine_tab(typename Gf2wTraits<LIMBS>::F f, Gf2wRecArgs<LIMBS> ra,
                                                                   typename Gf2wTraits<LIMBS>::E* __restrict__ out, size_t n) {
    typedef Gf2wTraits<LIMBS> Tr;
    typedef typename Tr::L L;
    constexpr int NPOS = 16 * LIMBS;
    constexpr int NW = 2 * LIMBS;                            // 32-bit words per element
    constexpr int SH = LIMBS == 2 ? 4 : 3;                   // log2(sizeof(L))
    constexpr uint32_t AMASK = 15u << SH;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    if constexpr (KT > 0) {
        L* T = reinterpret_cast<L*>(smem);                   // [KT][NPOS][16]
        for (int e = threadIdx.x; e < KT * NPOS * 16; e += REC_BLOCK) {
            const int j = e / (NPOS * 16), pos = (e / 16) % NPOS, v = e % 16;
            uint64_t clo = 0, chi = 0;
            if (4 * pos < 64) clo = (uint64_t)v << (4 * pos); else chi = (uint64_t)v << (4 * pos - 64);
            uint64_t rlo, rhi;
            if constexpr (LIMBS == 2) {
                // the constant may exceed degree n only if 4*pos+3 >= n: reduce it first
                u128e c;
                c.lo = clo;
                c.hi = chi;
                if (f.n < 128) c = f.reduce_raw(c);
                u128e lam;
                lam.lo = ra.lam_lo[j];
                lam.hi = ra.lam_hi[j];
                u128e r = f.mul(lam, c);
                rlo = r.lo;
                rhi = r.hi;
            } else {
                uint64_t c = f.n < 64 ? f.reduce_raw(clo) : clo;
                rlo = f.mul(ra.lam_lo[j], c);
                rhi = 0;
            }
            T[e] = Tr::pack(rlo, rhi);
        }
        __syncthreads();
    }
    // dynamic LDS starts at offset 0 of the workgroup's allocation (the kernel has no static LDS), so the table offsets
    // below are the look-ups' immediates; checked once, uniformly
    if (lds_offset_of(smem) != 0) __builtin_trap();
    constexpr uint32_t tbase = 0;
    const size_t gid = (size_t)blockIdx.x * REC_BLOCK + threadIdx.x;
    const size_t gsz = (size_t)gridDim.x * REC_BLOCK;
    for (size_t i = gid; i < n; i += gsz) {
        uint32_t acc[NW];
#pragma unroll
        for (int q = 0; q < NW; ++q) acc[q] = 0;
        // all operand loads of the element first (they stream; the look-ups below run in their shadow)
        uint32_t xw[KT > 0 ? KT : 1][NW];
#pragma unroll
        for (int j = 0; j < KT; ++j) {
            if constexpr (LIMBS == 2) {
                const uint4 x = ldg<true>(reinterpret_cast<const uint4*>(ra.trow[j]) + i);
                xw[j][0] = x.x; xw[j][1] = x.y; xw[j][2] = x.z; xw[j][3] = x.w;
            } else {
                const uint2 x = reinterpret_cast<const uint2*>(ra.trow[j])[i];
                xw[j][0] = x.x; xw[j][1] = x.y;
            }
        }
        // further rows of a group (coefficients that repeat): every load is issued before the first is consumed -- the
        // branches are scalar, and a load XORed inside its branch would wait there for its data, one HBM latency per row
        bool extra = false;
#pragma unroll
        for (int j = 0; j < KT; ++j) extra |= ra.gcnt[j] > 1;
        if (extra) {
            uint32_t ex[KT > 0 ? KT : 1][3][NW];
#pragma unroll
            for (int j = 0; j < KT; ++j)
#pragma unroll
                for (int r = 0; r < 3; ++r) {
#pragma unroll
                    for (int q = 0; q < NW; ++q) ex[j][r][q] = 0;
                    if (r + 1 < ra.gcnt[j]) {
                        if constexpr (LIMBS == 2) {
                            const uint4 x = ldg<true>(reinterpret_cast<const uint4*>(ra.erow[j][r]) + i);
                            ex[j][r][0] = x.x; ex[j][r][1] = x.y; ex[j][r][2] = x.z; ex[j][r][3] = x.w;
                        } else {
                            const uint2 x = reinterpret_cast<const uint2*>(ra.erow[j][r])[i];
                            ex[j][r][0] = x.x; ex[j][r][1] = x.y;
                        }
                    }
                }
#pragma unroll
            for (int j = 0; j < KT; ++j)
#pragma unroll
                for (int q = 0; q < NW; ++q) xw[j][q] ^= ff_xor3(ex[j][0][q], ex[j][1][q], ex[j][2][q]);
        }
        for (int p = 0; p < ra.kp; ++p) {                    // coefficient 1: XOR, no table (wave-uniform trip count)
            if constexpr (LIMBS == 2) {
                const uint4 x = ldg<true>(reinterpret_cast<const uint4*>(ra.prow[p]) + i);
                acc[0] ^= x.x; acc[1] ^= x.y; acc[2] ^= x.z; acc[3] ^= x.w;
            } else {
                const uint2 x = reinterpret_cast<const uint2*>(ra.prow[p])[i];
                acc[0] ^= x.x; acc[1] ^= x.y;
            }
        }
#pragma unroll
        for (int j = 0; j < KT; ++j) {
#pragma unroll
            for (int q0 = 0; q0 < NW; q0 += 2) {             // 16 look-ups (two words of the operand) per batch
                L t[16];
                // byte offset of entry v inside its position's 16-entry block = v << SH: mask the nibbles of a word in
                // place (one AND for the four high nibbles, shift + AND for the four low ones), then every look-up
                // address is ONE byte extraction; the block's own offset is the instruction's immediate
                uint32_t mh[2], ml[2];
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const uint32_t w = xw[j][q0 + e];
                    mh[e] = (SH == 4 ? w : (w >> (4 - SH))) & (AMASK * 0x01010101u);
                    ml[e] = (w << SH) & (AMASK * 0x01010101u);
                }
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    const int e = u >> 3, b = (u & 7) >> 1, hi = u & 1;
                    uint32_t a;                                  // byte b of the masked word: ONE v_bfe_u32 (the compiler
                    //                                              splits the intrinsic into shift + and)
                    asm("v_bfe_u32 %0, %1, %2, 8" : "=v"(a) : "v"(hi ? mh[e] : ml[e]), "n"(8 * b));
                    const int pos = 8 * (q0 + e) + 2 * b + hi;
                    t[u] = lds_entry<L>(tbase + (uint32_t)(((j * NPOS + pos) * 16 << SH) + a));
                }
#pragma unroll
                for (int u = 0; u < 16; u += 2) {
                    acc[0] = ff_xor3(acc[0], t[u].x, t[u + 1].x);
                    acc[1] = ff_xor3(acc[1], t[u].y, t[u + 1].y);
                    if constexpr (LIMBS == 2) {
                        acc[2] = ff_xor3(acc[2], t[u].z, t[u + 1].z);
                        acc[3] = ff_xor3(acc[3], t[u].w, t[u + 1].w);
                    }
                }
            }
        }
        if constexpr (LIMBS == 2) {
            stg<true>(reinterpret_cast<uint4*>(out) + i, make_uint4(acc[0], acc[1], acc[2], acc[3]));
        } else {
            reinterpret_cast<uint2*>(out)[i] = make_uint2(acc[0], acc[1]);
        }
    }
}

template <int LIMBS, int KT, bool DEEP>
static int launch_gf2w_rec(const void* policy, int device, const Gf2wRecArgs<LIMBS>& ra, void* out, size_t n, hipStream_t st) {
    typedef Gf2wTraits<LIMBS> Tr;
    const typename Tr::F& f = *reinterpret_cast<const typename Tr::F*>(policy);
    const size_t lds = (size_t)KT * 16 * LIMBS * 16 * sizeof(typename Tr::L);
    LaunchCfg lc = launch_cfg(device);
    size_t want = (n + REC_BLOCK - 1) / REC_BLOCK, cap;
    if (KT > 0) {
        // persistent grid: the table build (a few field multiplications per thread) is amortised over many elements
        int per_cu = (int)(160 * 1024 / (lds + 256));
        if (per_cu > 4) per_cu = 4;          // 4 x 512 threads = 32 waves: the CU's limit
        if (per_cu < 1) per_cu = 1;
        cap = (size_t)per_cu * (size_t)lc.num_cu;
    } else {
        cap = 0x7fffffff;                    // plain XOR of the rows: one element per thread, like the streaming kernels
    }
    const unsigned grid = (unsigned)(want < cap ? want : cap);
    if (lds > 48 * 1024) {
        static bool raised = false;     // per instantiation: allow more than the default dynamic LDS
        if (!raised) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gf2w_recombine_tab<LIMBS, KT, DEEP>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            raised = true;
        }
    }
    hipLaunchKernelGGL((k_gf2w_recombine_tab<LIMBS, KT, DEEP>), dim3(grid ? grid : 1), dim3(REC_BLOCK), lds, st, f, ra,
                       (typename Tr::E*)out, n);
    FFGPU_CHECK_LAUNCH();
    return 0;
}

template <int LIMBS>
static int dispatch_gf2w_rec(const void* policy, int device, const void* const* rows, const uint64_t* lam2, int k, void* out,
                             size_t n, hipStream_t st) {
    Gf2wRecArgs<LIMBS> ra;
    memset(&ra, 0, sizeof(ra));
    int kt = 0;                                                            // groups = tables
    bool used[REC_MAXK] = {false};
    for (int j = 0; j < k; ++j) {
        const uint64_t lo = lam2[2 * j], hi = lam2[2 * j + 1];
        if (used[j] || (lo == 0 && hi == 0)) continue;                     // coefficient 0: the row does not contribute
        if (lo == 1 && hi == 0) { ra.prow[ra.kp++] = rows[j]; continue; }  // coefficient 1: plain XOR
        ra.lam_lo[kt] = lo;
        ra.lam_hi[kt] = hi;
        ra.trow[kt] = rows[j];
        ra.gcnt[kt] = 1;
        used[j] = true;
        for (int j2 = j + 1; j2 < k && ra.gcnt[kt] < 4; ++j2)             // up to three more rows with this coefficient
            if (!used[j2] && lam2[2 * j2] == lo && lam2[2 * j2 + 1] == hi) {
                used[j2] = true;
                ra.erow[kt][ra.gcnt[kt] - 1] = rows[j2];
                ++ra.gcnt[kt];
            }
        ++kt;
    }
    if constexpr (LIMBS == 1) {
        // all coefficients 1 (the runtime's own for m = 3, 7) over 8-byte elements: the XOR of the rows is the same for
        // PAIRS of elements -- 16 bytes per lane and access, like every other streaming kernel here
        bool pairs = kt == 0 && ra.kp > 0 && n >= 2 && !(n & 1) && aligned16(out);
        for (int p = 0; p < ra.kp && pairs; ++p) pairs = aligned16(ra.prow[p]);
        if (pairs) {
            Gf2wRecArgs<2> r2;
            memset(&r2, 0, sizeof(r2));
            r2.kp = ra.kp;
            for (int p = 0; p < ra.kp; ++p) r2.prow[p] = ra.prow[p];
            GF2W128 unused;                                                    // (the policy is read by the table build only)
            memset(&unused, 0, sizeof(unused));
            return launch_gf2w_rec<2, 0, false>(&unused, device, r2, out, n / 2, st);
        }
    }
    // (DEEP = 16 look-ups per batch in flight for every table count > 0: the round-3 measurement; kt = 0 is a plain XOR)
#define GF2W_REC_CASE(KK)                                                                         \
    case KK:                                                                                      \
        return launch_gf2w_rec<LIMBS, KK, (KK > 0)>(policy, device, ra, out, n, st);
    switch (kt) {
        GF2W_REC_CASE(0) GF2W_REC_CASE(1) GF2W_REC_CASE(2) GF2W_REC_CASE(3) GF2W_REC_CASE(4) GF2W_REC_CASE(5)
        GF2W_REC_CASE(6) GF2W_REC_CASE(7) GF2W_REC_CASE(8) GF2W_REC_CASE(9)
        default: return 2;
    }
#undef GF2W_REC_CASE
}

// k rows (1..9), one output row; limbs selects GF2W64 / GF2W128.  Returns 2 if the shape is not covered.
int ffgpu_launch_gf2w_recombine(const void* policy, int limbs, int device, const void* const* rows, const uint64_t* lam2,
                                int k, void* out, size_t n, hipStream_t st) {
    if (k < 1 || k > REC_MAXK) return 2;
    return limbs == 2 ? dispatch_gf2w_rec<2>(policy, device, rows, lam2, k, out, n, st)
                      : dispatch_gf2w_rec<1>(policy, device, rows, lam2, k, out, n, st);
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

// ---- VALU issue-rate yardstick (the compute-side counterpart of k_copy16) --------------------------------------------
// What the integer VALU of THIS chip sustains at THIS moment, measured instead of assumed: every wave runs `iters` passes
// over 8 independent chains x 16 dependent instructions of one kind (inline asm: nothing is folded), the launch fills
// every SIMD with `waves_per_simd` waves.  One thread per launch also brackets its loop with s_memtime (shader cycles) and
// the constant 100 MHz counter, which gives the shader clock UNDER this load.  bench.py prices its VALU-bound rows
// (`valu_frac`) against these rates.
//   op 0: v_bitop3_b32 (the 3-input logic op of the GF(2^n) kernels)      op 1: v_add_u32      op 3: v_xor_b32
//   op 2: v_mad_u64_u32 (the 32 x 32 + 64 multiply-add every prime-field product is made of)     op 4: v_perm_b32
//   op 5: v_lshrrev_b32      op 6: v_and_or_b32      op 7: v_add3_u32      op 8: v_mul_lo_u32
//   op 9..11: v_alignbit_b32, v_lshl_or_b32, v_alignbyte_b32      op 12: v_lshrrev_b64      op 13: v_lshl_add_u64
// (measured, round 5: two-operand VOP2 instructions issue at ~2 cycles per wave64 once a SIMD holds two or more waves, the
// three-operand VOP3 ones and the multiplies at ~4: profiles/r05_valu_rates.md)
template <int OP>
__global__ __launch_bounds__(BLOCK) void k_valu_probe(uint32_t* __restrict__ sink, uint64_t* __restrict__ clk, int iters) {
    uint32_t a[8], b = threadIdx.x * 2654435761u + 1u, c = blockIdx.x * 40503u + 7u;
    uint64_t q[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        a[k] = b * (uint32_t)(k + 3);
        q[k] = ((uint64_t)a[k] << 32) | c;
    }
    const uint64_t t0 = __builtin_readcyclecounter();
    const uint64_t w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                if constexpr (OP == 0) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(a[k]) : "v"(b), "v"(c));
                else if constexpr (OP == 1) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[k]) : "v"(b));
                else if constexpr (OP == 2) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(q[k]) : "v"(b), "v"(c) : "vcc");
                else if constexpr (OP == 3) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a[k]) : "v"(b));
                else if constexpr (OP == 4) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c));
                else if constexpr (OP == 5) asm volatile("v_lshrrev_b32 %0, 1, %0" : "+v"(a[k]));
                else if constexpr (OP == 6) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c));
                else if constexpr (OP == 7) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c));
                else if constexpr (OP == 8) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[k]) : "v"(b));
                else if constexpr (OP == 9) asm volatile("v_alignbit_b32 %0, %0, %0, 7" : "+v"(a[k]));        // rotate (ChaCha)
                else if constexpr (OP == 10) asm volatile("v_lshl_or_b32 %0, %0, 12, %1" : "+v"(a[k]) : "v"(b));
                else if constexpr (OP == 11) asm volatile("v_alignbyte_b32 %0, %0, %0, 1" : "+v"(a[k]));
                else if constexpr (OP == 12) asm volatile("v_lshrrev_b64 %0, 3, %0" : "+v"(q[k]));            // 64-bit shift (carry passes)
                else asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(q[k]) : "v"(q[(k + 1) & 7]));          // 64-bit add
            }
        }
    }
    const uint64_t t1 = __builtin_readcyclecounter();
    const uint64_t w1 = wall_clock64();
    uint32_t x = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) x ^= a[k] ^ (uint32_t)q[k] ^ (uint32_t)(q[k] >> 32);
    if (x == 0x5a17c0deu) sink[0] = x;                         // keeps the chains alive; practically never true
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        clk[0] = t1 - t0;
        clk[1] = w1 - w0;
    }
}

// out[0] = lane-operations per second, out[1] = shader clock in MHz, out[2] = shader cycles per wave instruction and SIMD
int ffgpu_launch_valu_probe(int device, int op, int iters, int waves_per_simd, void* scratch32, double* out, hipStream_t st) {
    // SYNCHRONISES the stream (event + a blocking read-back of the cycle counts): a measurement aid, not capturable
    LaunchCfg lc = launch_cfg(device);
    if (op < 0 || op > 13 || iters < 1 || waves_per_simd < 1 || waves_per_simd > 8) return 1;
    const unsigned grid = (unsigned)(lc.num_cu * waves_per_simd);           // 256 threads = 4 waves = one per SIMD
    uint32_t* sink = (uint32_t*)scratch32;
    uint64_t* clk = (uint64_t*)((char*)scratch32 + 16);
    hipEvent_t e0, e1;
    if (hipEventCreate(&e0) != hipSuccess) return 1;
    if (hipEventCreate(&e1) != hipSuccess) {
        (void)hipEventDestroy(e0);
        return 1;
    }
    auto launch = [&](int n_it) {
#define FF_PROBE_CASE(OPV) case OPV: hipLaunchKernelGGL(k_valu_probe<OPV>, dim3(grid), dim3(BLOCK), 0, st, sink, clk, n_it); break;
        switch (op) { FF_PROBE_CASE(0) FF_PROBE_CASE(1) FF_PROBE_CASE(2) FF_PROBE_CASE(3) FF_PROBE_CASE(4) FF_PROBE_CASE(5)
                      FF_PROBE_CASE(6) FF_PROBE_CASE(7) FF_PROBE_CASE(8) FF_PROBE_CASE(9) FF_PROBE_CASE(10) FF_PROBE_CASE(11) FF_PROBE_CASE(12)
                      default: hipLaunchKernelGGL(k_valu_probe<13>, dim3(grid), dim3(BLOCK), 0, st, sink, clk, n_it); }
#undef FF_PROBE_CASE
    };
    launch(iters / 4 + 1);                                                    // warm-up: clocks ramp
    hipError_t err = hipEventRecord(e0, st);
    launch(iters);
    if (err == hipSuccess) err = hipGetLastError();
    if (err == hipSuccess) err = hipEventRecord(e1, st);
    if (err == hipSuccess) err = hipEventSynchronize(e1);
    float ms = 0.f;
    if (err == hipSuccess) err = hipEventElapsedTime(&ms, e0, e1);
    uint64_t host_clk[2] = {0, 0};
    if (err == hipSuccess) err = hipMemcpy(host_clk, clk, 16, hipMemcpyDeviceToHost);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    if (err != hipSuccess || ms <= 0.f) return 1;
    const double wave_instr = (double)grid * 4.0 * (double)iters * 128.0;
    out[0] = wave_instr * 64.0 / ((double)ms * 1e-3);
    out[1] = host_clk[1] ? (double)host_clk[0] / (double)host_clk[1] * 100.0 : 0.0;
    out[2] = (double)host_clk[0] / ((double)iters * 128.0 * (double)waves_per_simd);     // the timed wave shared its SIMD with the others
    return 0;
}
