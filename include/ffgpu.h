/*
 * ffgpu.h -- C ABI of libffgpu.so, the MI355X (gfx950) finite-field /
 * secret-sharing engine that sits behind MPyC's field-array and threshold
 * sharing interfaces.
 *
 * The reference (lschoe/mpyc) is pure Python and has no FFI of its own; the
 * seams this ABI plugs into are listed per entry point below as
 * "replaces: <reference file>:<lines>" (paths relative to the mpyc checkout).
 * INTEGRATION.md shows the ctypes binding a maintainer would add.
 *
 * Conventions
 *   - every entry point returns an int status (FFGPU_OK == 0); nothing throws
 *     across the boundary.  ffgpu_strerror() maps a status to text; the Python
 *     shim maps statuses to the exceptions the reference raises.
 *   - all array arguments are DEVICE pointers unless the name says host.
 *     Elements are fixed-width little-endian limbs in canonical form
 *     (0 <= x < modulus; for GF(2^n) the bit pattern of the polynomial):
 *         elem_bytes == 1   GF(2^n), n <= 8           uint8  [n]
 *         elem_bytes == 4   prime p < 2^32, GF(2^n) 9 <= n <= 32   uint32 [n]
 *         elem_bytes == 8   prime p < 2^64, GF(2^n) 33 <= n <= 64  uint64 [n]
 *         elem_bytes == 12  prime p = 2^k - c, 65 <= k <= 96, c < 2^31   3 x uint32 (12-byte LE integer)
 *         elem_bytes == 16  other primes p < 2^128, GF(2^n) n<=128 {lo,hi} uint64 pairs
 *         elem_bytes == 24  primes of 129..192 bits            3 x uint64 (24-byte LE integer, 8-byte aligned)
 *     This is exactly the byte layout of field.to_bytes() (finfields.py:91-102)
 *     for byte_length in {1,4,8,12,16,24} (shorter byte_lengths are zero-padded to the storage width).
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream).
 *     Calls are asynchronous on that stream; nothing synchronises unless the
 *     name says so.
 *   - inputs are never written; `out` may alias an input of the same shape for
 *     the element-wise entry points (in-place operators, finfields.py:1068-1124).
 *   - thread-safety: the field description of a context is immutable after
 *     creation and a context may be shared between host threads and streams;
 *     its only mutable state -- the ffgpu_matmul scratch buffers (one per
 *     stream, under a mutex) and the lazily built tables -- is internally
 *     synchronised.  Two host threads must not issue work on the SAME stream
 *     of a context concurrently.  The reference only ever calls from one
 *     event-loop thread (asyncoro.py:416-464).
 */
#ifndef FFGPU_H
#define FFGPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FFGPU_ABI_VERSION 1

/* status codes */
#define FFGPU_OK          0
#define FFGPU_EINVAL      1   /* bad argument (null pointer, t >= m, k == 0, ...)   */
#define FFGPU_ENOTSUP     2   /* field / size outside what this build supports      */
#define FFGPU_EHIP        3   /* a HIP runtime call failed; see ffgpu_last_hip_error */
#define FFGPU_EMODULUS    4   /* modulus is not usable (even/zero/one, too wide)     */
#define FFGPU_ENOMEM      5
#define FFGPU_ESTALE      6   /* ffgpu_ipc_read: the mapping does not show the row the descriptor describes */

/* field kinds */
#define FFGPU_PRIME   1   /* GF(p), p prime < 2^192       (finfields.py:347-363 pGF)  */
#define FFGPU_BINARY  2   /* GF(2^n), 1 <= n <= 128       (finfields.py:508-525 xGF,
                             gfpx.py:848-1121 BinaryPolynomial)                       */

/* reduction strategy picked at context creation (reported for logs/tests) */
#define FFGPU_RED_PSEUDO_MERSENNE 1  /* p = 2^k - c, small c: fold reduction          */
#define FFGPU_RED_RECIPROCAL      2  /* arbitrary modulus: Barrett-type reciprocal    */
#define FFGPU_RED_GF2_SWAR        3  /* GF(2^n), n<=8: packed shift-xor               */
#define FFGPU_RED_GF2_WIDE        4  /* GF(2^n), n<=128: limb shift-xor               */
#define FFGPU_RED_MONTGOMERY      5  /* arbitrary odd 65..192-bit p: word-serial REDC */

typedef struct ffgpu_ctx ffgpu_ctx;

/* ---- library / device ------------------------------------------------- */
int         ffgpu_abi_version(void);
const char* ffgpu_strerror(int status);
const char* ffgpu_last_hip_error(void);          /* text of the last failing HIP call */
int         ffgpu_device_count(int* count);
/* PCI bus id ("0000:05:00.0") of a device into buf (len >= 16): lets a multi-process run show that its ranks sit on
 * distinct GPUs.  No reference counterpart (the reference has no device).                                          */
int         ffgpu_device_pci_bus_id(int device, char* buf, int len);

/* ---- field context ---------------------------------------------------- */
/* modulus: little-endian uint64 limbs.  FFGPU_PRIME: the prime p, nlimbs 1..3: primes of 129..192 bits are stored
 * as three 64-bit limbs = 24 bytes per element (p = 2^k - c with c < 2^31 -- the l+32-bit default fields of
 * SecInt(97..160), sectypes.py:673-676 -- reduce by folding, any other odd prime of that size -- e.g. the
 * root-of-unity primes of SecInt(l, n=N), finfields.py:332-343 -- by Montgomery products); above 192 bits:
 * FFGPU_ENOTSUP.
 * FFGPU_BINARY: bit pattern of the irreducible polynomial including its leading
 * term (degree n <= 128 needs up to 3 limbs).  Primality / irreducibility is the
 * caller's job (the reference checks it in pGF/xGF before any array exists).
 * replaces: finfields.py:23-60 (GF / arrayGF pick the array representation). */
int ffgpu_ctx_create(int kind, const uint64_t* modulus, int nlimbs, int device,
                     ffgpu_ctx** out);
int ffgpu_ctx_destroy(ffgpu_ctx* ctx);
/* Opt-in timing: with enable != 0 every compute call records a pair of events on its stream around the
 * kernels it launches; ffgpu_last_kernel_ms waits for the most recent call and returns its GPU time
 * (FFGPU_EINVAL if timing is off or nothing has been launched).  Off by default: no events, no cost.   */
int ffgpu_ctx_set_timing(ffgpu_ctx* ctx, int enable);
int ffgpu_last_kernel_ms(ffgpu_ctx* ctx, float* ms);
/* enable == 2: ACCUMULATE -- every compute call keeps its own event pair; ffgpu_busy_ms waits for the calls issued so
 * far and returns the summed GPU time of all calls since the last reset (and their number): the "GPU busy" numerator
 * of an API-level measurement (bench.py `api`).  Calls on one stream do not overlap, so the sum is busy time.        */
int ffgpu_busy_ms(ffgpu_ctx* ctx, double* ms, unsigned long long* calls, int reset);
int ffgpu_ctx_elem_bytes(const ffgpu_ctx* ctx);   /* 1, 4, 8, 12, 16 or 24       */
/* Host SCALARS (constants, Lagrange coefficients, matrix entries, PRSS weights: every `const uint64_t* host_...`
 * argument that is documented as "canonical 2-limb scalars") occupy ffgpu_ctx_scalar_limbs(ctx) little-endian
 * limbs each: 2, or 3 for the 24-byte prime fields.  Exponents (ffgpu_pow) may have up to 3 limbs.            */
int ffgpu_ctx_scalar_limbs(const ffgpu_ctx* ctx);
int ffgpu_ctx_reduction(const ffgpu_ctx* ctx);    /* one of FFGPU_RED_*          */
int ffgpu_ctx_device(const ffgpu_ctx* ctx);

/* ---- device memory (optional: any hipMalloc'ed / torch pointer works) -- */
int ffgpu_malloc(ffgpu_ctx* ctx, size_t bytes, void** dptr);
int ffgpu_free(ffgpu_ctx* ctx, void* dptr);
int ffgpu_h2d(ffgpu_ctx* ctx, void* dst, const void* host_src, size_t bytes, void* stream);
int ffgpu_d2h(ffgpu_ctx* ctx, void* host_dst, const void* src, size_t bytes, void* stream);
int ffgpu_stream_sync(ffgpu_ctx* ctx, void* stream);

/* ---- canonical reduction ---------------------------------------------- */
/* out[i] = raw[i] mod modulus for arbitrary limb patterns of elem_bytes width.
 * replaces: finfields.py:717-725 (`value %= modulus` in FiniteFieldArray.__init__) */
int ffgpu_reduce(ffgpu_ctx* ctx, const void* raw, void* out, size_t n, void* stream);

/* ---- element-wise field arithmetic, arrays of equal length n ---------- */
/* replaces: finfields.py:1056-1103 (__add__/__sub__/__iadd__/__isub__),
 *           :1105-1124 (__mul__/__imul__), :1189-1192 (__neg__);
 *           GF(2^n): gfpx.py:982-1045 (_add/_mul/_mod).                      */
int ffgpu_add(ffgpu_ctx* ctx, const void* a, const void* b, void* out, size_t n, void* stream);
int ffgpu_sub(ffgpu_ctx* ctx, const void* a, const void* b, void* out, size_t n, void* stream);
int ffgpu_mul(ffgpu_ctx* ctx, const void* a, const void* b, void* out, size_t n, void* stream);
int ffgpu_neg(ffgpu_ctx* ctx, const void* a, void* out, size_t n, void* stream);

/* array (op) scalar; scalar is a HOST value in canonical limbs (2 x uint64,
 * little-endian, upper limb 0 for narrow fields).
 * replaces: the `other` is int / field element branch of finfields.py:1045-1054. */
int ffgpu_add_scalar(ffgpu_ctx* ctx, const void* a, const uint64_t* host_scalar, void* out, size_t n, void* stream);
int ffgpu_mul_scalar(ffgpu_ctx* ctx, const void* a, const uint64_t* host_scalar, void* out, size_t n, void* stream);
/* out = scalar - a   (finfields.py:1084-1091 __rsub__)                        */
int ffgpu_rsub_scalar(ffgpu_ctx* ctx, const void* a, const uint64_t* host_scalar, void* out, size_t n, void* stream);

/* out = a * b + c   (fused multiply-add; one pass instead of two)            */
int ffgpu_muladd(ffgpu_ctx* ctx, const void* a, const void* b, const void* c, void* out, size_t n, void* stream);

/* ---- second-tier element-wise operations -------------------------------- */
/* out[i] = a[i]^e for a public exponent e >= 0 (little-endian uint64 limbs, exp_limbs <= 2).
 * One pass: square-and-multiply in registers.
 * replaces: finfields.py:1159-1187 (__pow__), :1408-1414 (_pow via gmpy2.powmod per element);
 * sqrt for p = 3 mod 4 is pow by (p+1)/4 (:1424-1458), Legendre symbol pow by (p-1)/2 (:1460-1470). */
int ffgpu_pow(ffgpu_ctx* ctx, const void* a, const uint64_t* host_exp, int exp_limbs, void* out, size_t n,
              void* stream);
/* out[i] = a[i]^-1.  Batched with Montgomery's trick inside each thread (3 products per element
 * plus one exponentiation per 16..128 elements).  a[i] == 0 gives out[i] = 0 and ORs 1 into
 * *dev_zero_flag (device int32, may be NULL): the reference raises ZeroDivisionError there
 * (gmpy.py:197-210), which the host wrapper does after reading the flag.
 * replaces: finfields.py:1278-1281 (reciprocal), :1416-1422 (_reciprocal via gmpy2.invert per element). */
int ffgpu_inv(ffgpu_ctx* ctx, const void* a, void* out, size_t n, void* dev_zero_flag, void* stream);

/* ---- Beaver-triple combination (NOT a reference function; parity UNPINNED) ------------- */
/* out = z + d*y + e*x (+ d*e if add_de != 0), element-wise: the local step of Beaver multiplication
 * after d = a - x and e = b - y have been opened ([x],[y],[z] = shares of a triple z = x*y).  add_de:
 * whether this party adds the public term d*e -- every party under Shamir sharing (a public constant is
 * shared by the constant polynomial), exactly one party under additive sharing.  MPyC itself multiplies with GRR resharing (runtime.py:603-689) and
 * has no Beaver triples; this entry point exists because the project brief names
 * it and is validated only against the textbook identity, not against reference outputs.          */
int ffgpu_beaver_combine(ffgpu_ctx* ctx, const void* z, const void* x, const void* y, const void* d,
                         const void* e, int add_de, void* out, size_t n, void* stream);

/* ---- Shamir share generation ------------------------------------------ */
/* shares[i][h] = secrets[h] + sum_{j<t} coeffs[j][h] * (i+1)^(j+1)  (mod modulus),
 * i = 0..m-1, h = 0..n-1;  coeffs is (t, n) row-major with row stride
 * coeff_stride elements, shares is (m, n) row-major with row stride
 * share_stride elements (strides let callers keep rows 16-byte aligned).
 * For GF(2^n) the x-coordinate of party i is the polynomial with bit pattern i+1.
 * This is the np-path coefficient convention: coeffs[j] multiplies X^(j+1).
 * replaces: thresha.py:47-64 np_random_split (C drawn at :60, V @ [s;C] at :61-63).
 * The list path thresha.py:23-44 random_split is the same map with
 * coeffs[j][h] = c_h[t-1-j].                              */
int ffgpu_split(ffgpu_ctx* ctx, const void* secrets, const void* coeffs, size_t coeff_stride,
                int t, int m, void* shares, size_t share_stride, size_t n, void* stream);

/* Fused local product + share generation: secrets[h] = a[h]*b[h] is never
 * written to memory.
 * replaces: runtime.py:1134-1138 (c = a * b; c = self._reshare(c)) up to the
 * send at runtime.py:661-667.                                                  */
int ffgpu_mul_split(ffgpu_ctx* ctx, const void* a, const void* b, const void* coeffs,
                    size_t coeff_stride, int t, int m, void* shares, size_t share_stride,
                    size_t n, void* stream);

/* ---- share generation with the on-device CSPRNG ------------------------- */
/* Same maps as ffgpu_split / ffgpu_mul_split, but the t*n coefficients are drawn inside the
 * kernel from a ChaCha keystream (host_key32: 32 bytes from the host CSPRNG, fresh per call or
 * with a fresh nonce; rounds: 20, 12 or 8; 0 = 20) and never touch HBM.  The keystream layout and
 * the 2^-64-bias field sampler are documented in mpyc_amd/csrc/rng.hpp; ffgpu_rng_coeffs writes
 * exactly the coefficient matrix the fused kernels consume for the same (key, nonce, rounds, t),
 * so split_rng(key) == split(rng_coeffs(key)) bit for bit.
 * replaces: the secrets.randbelow draws of thresha.py:37 and :58-60 (one OS-CSPRNG call per
 * coefficient in the reference).                                                            */
int ffgpu_rng_coeffs(ffgpu_ctx* ctx, const uint8_t* host_key32, uint64_t nonce, int rounds, int t,
                     void* coeffs, size_t coeff_stride, size_t n, void* stream);
int ffgpu_split_rng(ffgpu_ctx* ctx, const void* secrets, const uint8_t* host_key32, uint64_t nonce,
                    int rounds, int t, int m, void* shares, size_t share_stride, size_t n, void* stream);
int ffgpu_mul_split_rng(ffgpu_ctx* ctx, const void* a, const void* b, const uint8_t* host_key32,
                        uint64_t nonce, int rounds, int t, int m, void* shares, size_t share_stride,
                        size_t n, void* stream);

/* Fused chain gate: shares[i][h] = share of (A[h] * B[h]) for party i+1 where BOTH factors are given as
 * recombinations, A[h] = sum_j lambda_a[j] * rows_a[j][h] (ka <= 7 rows) and B likewise (kb = 0: B = A, a
 * squaring; a factor that already exists as an array is the 1-row case with lambda = 1).  One pass: recombine
 * in registers, multiply, draw the degree-t coefficients from the device CSPRNG (dev_state != NULL: the
 * device-resident state, else host key / nonce / rounds), write the m share rows.  In a chain of secure
 * multiplications this is `np_recombine` of the previous gate (thresha.py:119-132) + the local product and
 * `np_random_split` of the next one (runtime.py:1134-1138, thresha.py:47-64) without the recombined share
 * ever going to HBM: 6 instead of 9 memory accesses per element for a squaring.  t <= 3, else FFGPU_ENOTSUP
 * (callers then use ffgpu_recombine + ffgpu_mul_split_rng).  lambda: canonical 2-limb host scalars.          */
int ffgpu_gate_rng(ffgpu_ctx* ctx, const void* const* host_rows_a, const uint64_t* host_lambda_a, int ka,
                   const void* const* host_rows_b, const uint64_t* host_lambda_b, int kb, const uint8_t* host_key32,
                   uint64_t nonce, int rounds, void* dev_state, int t, int m, void* shares, size_t share_stride,
                   size_t n, void* stream);

/* `nbatch` independent chain gates in ONE launch (grid rows): gate y reads every row of A at element offset
 * y * batch_stride_a (B: y * batch_stride_b) from the given row pointers and writes its m share rows at
 * shares + y * batch_stride_out (+ i * share_stride for party i+1); its coefficients come from the call's
 * generator stream with y added to bits 8..15 of nonce word 1 (nbatch <= 255), so the gates draw independent
 * randomness (dev_state / nonce / defer_advance: see ffgpu_rng_state_advance below).  Bits 40..47 of `nonce` are
 * therefore RESERVED for the batch row on the host-key path: nbatch > 1 requires nonce < 2^40 (FFGPU_EINVAL
 * otherwise), so that (key, nonce, row) never aliases (key, nonce', row') under a reused key.  This is what the 2t+1 re-sharing parties of one `_reshare` (runtime.py:658-666) do when a whole
 * computation is held on one GPU: with the sub-shares stored [recipient][sender][n] the senders of the next gate
 * are batch_stride = (2t+1)*n apart and a layer of the np_aes S-box chain (runtime.py:1356-1367) is one launch
 * instead of 2t+1.  nbatch = 1 is ffgpu_gate_rng.                                                              */
int ffgpu_gate_rng_batch(ffgpu_ctx* ctx, const void* const* host_rows_a, const uint64_t* host_lambda_a, int ka,
                         size_t batch_stride_a, const void* const* host_rows_b, const uint64_t* host_lambda_b, int kb,
                         size_t batch_stride_b, const uint8_t* host_key32, uint64_t nonce, int rounds, void* dev_state,
                         int defer_advance, int t, int m, void* shares, size_t share_stride, size_t batch_stride_out,
                         size_t n, int nbatch, void* stream);

/* Deferred advance of a device-resident generator state: with dev_state != NULL, ffgpu_gate_rng_batch draws from
 * (state nonce + `nonce`) -- `nonce` < 2^32 is then an OFFSET -- and, if defer_advance != 0, leaves the state alone.
 * A sequence of N launches uses offsets 0..N-1 and ends with ONE ffgpu_rng_state_advance(state, N), instead of one
 * nonce update (a one-thread kernel for large grids, ~4 us) after every launch; captured in a HIP graph the offsets
 * are constants and the single advance keeps every replay on fresh nonces.                                        */
int ffgpu_rng_state_advance(ffgpu_ctx* ctx, void* dev_state, uint32_t by, void* stream);

/* Device-resident generator state, for launches captured in a HIP graph: the kernels read key / nonce /
 * rounds from `dev_state` (ffgpu_rng_state_bytes() bytes of device memory) when they start, and the nonce is
 * advanced on the device after every use (by the last workgroup of the share-generation kernel itself), so each REPLAY of a captured ffgpu_split_rng_state draws fresh
 * coefficients (a host key in the kernel arguments would be frozen into the graph).  One state per stream of
 * launches; ffgpu_rng_state_init is not capturable (it synchronises).  mul_by: NULL or the second factor of
 * the fused local product (as ffgpu_mul_split_rng).  No reference counterpart (secrets.randbelow, thresha.py:58). */
size_t ffgpu_rng_state_bytes(void);
int ffgpu_rng_state_init(ffgpu_ctx* ctx, void* dev_state, const uint8_t* host_key32, uint64_t nonce, int rounds,
                         void* stream);
int ffgpu_split_rng_state(ffgpu_ctx* ctx, const void* secrets, const void* mul_by, void* dev_state, int t, int m,
                          void* shares, size_t share_stride, size_t n, void* stream);

/* ---- Lagrange recombination ------------------------------------------- */
/* out[r][h] = sum_{j<k} lambda[r][j] * rows[j][h]  (mod modulus), r < w.
 * host_rows: HOST array of k device pointers (rows arrive from k peers and need
 * not be contiguous).  host_lambda: HOST array (w, k) of canonical scalars,
 * 2 x uint64 limbs each (from _recombination_vector, thresha.py:67-85, computed
 * on the host and cached there).  out is (w, n) with row stride out_stride.
 * Products are accumulated unreduced and reduced once, as the reference's
 * object matmul does (finfields.py:1126-1135).
 * replaces: thresha.py:119-132 np_recombine, thresha.py:88-116 recombine.     */
int ffgpu_recombine(ffgpu_ctx* ctx, const void* const* host_rows, const uint64_t* host_lambda,
                    int k, int w, void* out, size_t out_stride, size_t n, void* stream);

/* ---- dense matrix product ---------------------------------------------- */
/* C (M x N) = A (M x K) @ B (K x N) over the field, row-major with leading dimensions lda/ldb/ldc in
 * ELEMENTS; C must not overlap A or B.  The result is that of the reference's object matmul followed by one `%`
 * (exact integer accumulation, reduced at the end).  Kernel families, chosen by shape and field: one output dimension <= 8 ->
 * matrix x few columns / few rows x matrix kernels that read the big operand once (one-word primes: six multiply-adds per
 * term into column sums); prime fields with more than 8 rows and columns, K >= 64 and M*N*K >= 8e7 -> exact signed-digit
 * GEMMs on the int8 matrix cores (tiles padded to 64); otherwise an LDS-tiled vector-ALU kernel (split over K when
 * the output has few tiles).  The matrix-core and split-K paths keep digit planes / partial sums in grow-only
 * scratch buffers owned by the context, ONE PER STREAM (launches on different streams never share scratch): the
 * first call of a larger shape on a stream allocates its buffer (and synchronises that stream), so issue one such
 * call before capturing launches in a HIP graph.
 * replaces: finfields.py:1126-1146 (__matmul__: object matmul then one `%`), the local product of
 * runtime.py:2481-2541 np_matmul (A @ B at :2531).                                             */
int ffgpu_matmul(ffgpu_ctx* ctx, const void* A, size_t lda, const void* B, size_t ldb, void* C, size_t ldc,
                 size_t M, size_t K, size_t N, void* stream);

/* ---- square roots, p = 1 (mod 4) --------------------------------------------- */
/* out[i] = the square root the reference returns for a[i] (Cipolla-Lehmer with the smallest b such that
 * b^2 - 4a is a non-residue; 0 for a = 0).  Primes p = 3 (mod 4) and GF(2^n) take ffgpu_pow with the
 * exponents (p+1)/4 resp. q/2; FFGPU_ENOTSUP for those here.
 * replaces: finfields.py:447-470 (PrimeFieldElement._sqrt) mapped over arrays by :1459-1460.          */
int ffgpu_sqrt_cl(ffgpu_ctx* ctx, const void* a, void* out, size_t n, void* stream);

/* ---- Gaussian elimination ---------------------------------------------------- */
/* In place on `batch` row-major (n x ncols) matrices stored back to back (ncols >= n).
 * mode 0 (solve): (A | B) -> (. | A^-1 B), Gauss-Jordan; the solution occupies columns n..ncols-1.
 * mode 1 (det):   forward elimination only; det_out[b] = product of the pivots chosen by the reference's
 *                 rule (first nonzero entry at or below the diagonal; finfields.py:933-947 -- NB the
 *                 reference does not negate on row swaps, and neither does this), 0 if singular.
 * dev_singular: `batch` ints in device memory, set to 1 for singular matrices (whose contents are then
 *               unspecified); the caller raises ZeroDivisionError('no inverse exists') (finfields.py:893).
 * replaces: finfields.py:872-908 gauss_solve, :910-916 gauss_inv, :918-955 gauss_det
 *           (np.linalg.solve / inv / det / matrix_power with negative exponent on field arrays).     */
int ffgpu_gauss(ffgpu_ctx* ctx, void* a, int n, int ncols, size_t batch, int mode, void* det_out,
                void* dev_singular, void* stream);

/* ---- small public matrix over the last axis --------------------------------- */
/* out[i*r + a] = bias[a] + sum_{c<g} M[a][c] * in[i*g + c]  for every group i of g consecutive elements
 * (r, g <= 16).  host_matrix: (r, g) canonical 2-limb scalars; host_bias: r scalars or NULL.
 * replaces: finfields.py:1126-1146 `A @ x[..., np.newaxis]` with a public A and trailing axis g
 * (demos/np_aes.py:40-41: the S-box's GF(2) affine map on the 8 bit-shares of every byte, `+ B`),
 * and runtime.py:4475-4484 np_from_bits (sum_j x_j * 2^j over the last axis: r = 1).               */
int ffgpu_group_matvec(ffgpu_ctx* ctx, const uint64_t* host_matrix, const uint64_t* host_bias, int r, int g,
                       const void* in, void* out, size_t ngroups, void* stream);

/* ---- reductions ------------------------------------------------------------- */
/* out[0] = sum_i a[i]*b[i]  (ffgpu_dot)  /  sum_i a[i]  (ffgpu_sum), one field element.
 * workspace: device scratch of at least FFGPU_REDUCE_WORKSPACE_BYTES bytes (per concurrent call).
 * replaces: the local part of runtime.py in_prod / np_sum-style reductions (sum(map(mul, x, y)) on
 * shares, then one reshare) and FiniteFieldArray reductions through __array_function__ (np.sum).    */
#define FFGPU_REDUCE_WORKSPACE_BYTES (1024 * 16)
int ffgpu_dot(ffgpu_ctx* ctx, const void* a, const void* b, void* out, void* workspace, size_t n, void* stream);
int ffgpu_sum(ffgpu_ctx* ctx, const void* a, void* out, void* workspace, size_t n, void* stream);

/* ---- pseudorandom secret sharing: combination step ----------------------- */
/* HOST function: out_i = SHAKE128(msg_i) squeezed to out_len bytes, for nstreams independent messages on up
 * to `threads` host threads (<= 0: all cores).  One sponge is sequential, so a stream cannot be spread over
 * GPU lanes; the C(m, t) subset keys of a PRSS call can be spread over host cores.  Outputs go to caller
 * buffers (pin them and upload them as the `host_streams` of ffgpu_prss_combine).
 * replaces: thresha.py:255 `shake_128(self.key + s).digest(n * self.byte_length)`, once per key.         */
/* 1 if ffgpu_shake128_expand squeezes through the system's libcrypto (loaded at run time, checked against the
 * SHAKE128("") known answer), 0 if through the portable Keccak-f[1600] of the library (also with FFGPU_SHAKE_OWN=1). */
int ffgpu_shake128_backend(void);
int ffgpu_shake128_expand(const uint8_t* const* msgs, const size_t* msg_lens, int nstreams, size_t out_len,
                          uint8_t* const* outs, int threads);
/* The same streams, RESUMABLE: open absorbs the nstreams messages (*handle owns the sponge states), every squeeze
 * call writes the NEXT nbytes of stream i to outs[i] (nstreams host pointers; up to `threads` host threads, <= 0: all
 * cores), close frees the handle.  This is how the mirror runs a large PRSS call: a slice of every stream is squeezed
 * into one half of a bounded pinned buffer while the previous slice uploads and ffgpu_prss_combine consumes it on the
 * device (measured on the MI355X box, m = 7, t = 3, 10^7 draws: profiles/r03_api_path.md).
 * replaces: the same line, thresha.py:255 (hashlib squeezes a stream in one piece).                                */
int ffgpu_shake128_open(const uint8_t* const* msgs, const size_t* msg_lens, int nstreams, void** handle);
int ffgpu_shake128_squeeze(void* handle, uint8_t* const* outs, size_t nbytes, int threads);
void ffgpu_shake128_close(void* handle);

/* out[h] (+)= sum_{s<ks} sum_{j<d} draw_s[h*d + j] * weights[s][j]   (mod modulus)
 * host_streams: HOST array of ks DEVICE pointers to the raw SHAKE128 output of subset s
 * (n*d*l bytes each; the XOF is sequential per key and is computed on the host with hashlib, as in
 * thresha.PRF.__call__, thresha.py:238-266).  draw = l-byte little-endian chunk reduced into
 * range(bound): mask_bits == 0 -> bound is the field order (wide reduction, l = byte_length + 16),
 * mask_bits = b > 0 -> bound = 2^b (runtime.py:4062-4076 rounds bounds to powers of two).
 * host_weights: (ks, d) canonical 2-limb scalars f_S(i) * (i+1)^power from _f_S_i (thresha.py:135-141).
 * replaces: thresha.py:163-173 np_pseudorandom_share (d = 1), :201-217 np_pseudorandom_share_0,
 * and the list versions :144-160, :176-198.                                                    */
int ffgpu_prss_combine(ffgpu_ctx* ctx, const void* const* host_streams, int ks, int d, int l, int mask_bits,
                       const uint64_t* host_weights, int accumulate, void* out, size_t n, void* stream);

/* PRODUCTION mode of the same combination: the draws come from a COUNTER-MODE PRF expanded on the device instead of the
 * reference's SHAKE128 stream (a sponge is sequential per key: in parity mode the host squeezes and the device idles).
 * host_keys: ks x 40 bytes, per subset key the 32-byte ChaCha key followed by the 8-byte nonce, both derived by the caller
 * from (PRF key, common input) -- the mirror uses shake_128(b"mpyc_amd prss chacha v1\0" + len(key) + key + input).digest(40).
 * A draw is l little-endian KEYSTREAM bytes reduced into range(bound) by the reference's own rule (`% bound`,
 * l = byte_length + len(key), thresha.py:234-236, or a mask for bound = 2^mask_bits), so a draw depends on (key, input,
 * index, bound) only -- not on the field (runtime.py:758-761 evaluates one set of PRFs over two fields).  rounds: 20, 12 or 8.
 * Layout of the keystream: ffgpu_prss_chacha_layout (tb blocks per tile of dpt draws, ceil(l/4) words per draw; draw j of
 * element h sits in tile h / dpt, slot h % dpt, block counters (tile * d + j) * tb + b).  Every party must use the same
 * mode; there is no reference counterpart for the PRF itself (pinned to RFC 8439 and to oracle/fforacle.c
 * orc_prss_chacha, not to reference outputs), the combination is the reference's.  Limits per call (FFGPU_EINVAL beyond
 * them): ks <= 32, ks * d <= 64, 1 <= l <= 64 bytes per draw (the mirror raises NotImplementedError for wider draws).
 * replaces: thresha.py:163-173, 201-217 (np_pseudorandom_share, np_pseudorandom_share_0) with PRF := ChaCha.      */
int ffgpu_prss_chacha(ffgpu_ctx* ctx, const uint8_t* host_keys, int ks, int d, int l, int mask_bits, int rounds,
                      const uint64_t* host_weights, int accumulate, void* out, size_t n, void* stream);
int ffgpu_prss_chacha_layout(int l, int* tb, int* dpt);

/* ---- GF(2^8) S-box layer (local / public values) ----------------------- */
/* GF(2^n<=8), the linear layer of the AES S-box on BIT SHARES: for every group of 8 elements (the shares of the
 * 8 bits of one byte, 8-byte aligned): y = M x + bias with a public 8x8 matrix M (row-major canonical scalars);
 * from_bits = 0: out = y (8 elements per group); from_bits = 1: out = sum_r 2^r y_r (one element per group).
 * replaces: demos/np_aes.py:40-42 (`A @ x[..., np.newaxis]`, `x += B`, `mpc.np_from_bits(x)`) in one pass;
 *           finfields.py:1126-1146 + runtime.py:4475-4484.  ffgpu_group_matvec is the general form.           */
int ffgpu_gf256_bit_affine(ffgpu_ctx* ctx, const uint64_t* host_matrix, const uint64_t* host_bias, int from_bits,
                           const void* in, void* out, size_t n, void* stream);

/* GF(2^n<=8): out[8 i + j] = ((in[i] >> j) & 1) + addend[8 i + j] for PUBLIC bytes `in` (addend may be NULL):
 * the local tail of the secure bit decomposition over a binary field, where every party adds the bits of the
 * opened masked value c to its shares of the random bits.
 * replaces: runtime.py:4418-4423 (`c_bits = np.int8(np.right_shift.outer(c, shifts) & 1); return c_bits + r_bits`). */
int ffgpu_gf256_to_bits(ffgpu_ctx* ctx, const void* in, const void* addend, void* out, size_t n, void* stream);

/* The secure bit decomposition of runtime.np_to_bits over GF(2^8) (runtime.py:4411-4423) with the two local steps
 * around its opening fused, for a computation whose parties all live on this GPU (mpyc_amd/protocols.py):
 *
 * ffgpu_gf256_mask_open: out[h] = sum_r coef[r] * rows[r][h] + sum_p mu[p] * (sum_b 2^b rbits_p[8h + b])
 *   = the opened masked value c = a + r_modl (runtime.py:4414-4421: `r_modl` by Horner over the bit shares, `a + r_modl`,
 *   `output`) computed straight from the t+1 opening parties' shares: rows/coef = their shares of a, each possibly
 *   still a pending recombination of sub-share rows (coef = mu_p * lambda_s, <= 32 rows), rbits_p = their shares of the
 *   random bits (8 per byte, 16-byte aligned, <= 8 parties).  One pass instead of 3 per party + 1.
 * ffgpu_gf256_bits_affine_fold: out_y[h] = sum_r 2^r (M (bits(c[h]) + rbits_y[8h..8h+7]) + bias)_r for parties
 *   y < nbatch (rbits_y = rbits + y*rbits_batch_stride, out_y = out + y*out_batch_stride; one launch, grid rows = y)
 *   = `c_bits + r_bits` (runtime.py:4422-4423), the public affine map over the bit shares and np_from_bits
 *   (demos/np_aes.py:40-42, runtime.py:4475-4484) without the 8n-byte bit arrays ever reaching HBM.                */
int ffgpu_gf256_mask_open(ffgpu_ctx* ctx, const void* const* host_rows, const uint64_t* host_coef, int nrows,
                          const void* const* host_rbits, const uint64_t* host_mu, int np, void* out, size_t n, void* stream);
int ffgpu_gf256_bits_affine_fold(ffgpu_ctx* ctx, const uint64_t* host_matrix, const uint64_t* host_bias, const void* c,
                                 const void* rbits, size_t rbits_batch_stride, void* out, size_t out_batch_stride, size_t n,
                                 int nbatch, void* stream);

/* The WHOLE secure AES S-box layer (demos/np_aes.py:37-43) for all m parties of a computation held on one GPU, in ONE
 * launch: x^254 by the reference's addition chain (11 secure multiplications: local products of the 2t+1 senders,
 * re-sharing with coefficients from the device CSPRNG, recombination with host_lambda = the Lagrange vector of the
 * senders 1..2t+1 at 0; runtime.py:1356-1367, 1096-1141, 603-689), np_to_bits (opening of y + r_modl from parties
 * 1..t+1 with host_mu, bits + r_bits; runtime.py:4411-4423), the GF(2) affine map host_matrix / host_bias on the bit
 * shares and np_from_bits.  Everything is element-wise, so a thread carries the m shares of four bytes through the
 * whole protocol in registers: 10 m bytes of HBM traffic per secure byte instead of the 269 of the per-step kernels.
 * x / out: m rows of n bytes (row strides in bytes, multiples of 4); rbits: m rows of 8 n bit shares (stride a
 * multiple of 16).  GF(2^8) only; any n (the n % 4 bytes after the last whole word of each row are handled with byte
 * accesses), every (m, t) with 2t+1 <= m <= 7, t <= 3 that the reference's defaults and t = 1 produce: (3,1), (4,1),
 * (5,1), (5,2), (6,1), (6,2), (7,1), (7,2), (7,3) -- FFGPU_ENOTSUP otherwise, and for unaligned rows (compose the
 * layer from ffgpu_gate_rng_batch / ffgpu_gf256_mask_open / ffgpu_gf256_bits_affine_fold then).  Randomness: as
 * ffgpu_gate_rng_batch (host key + nonce + rounds, or dev_state with `nonce` as offset and defer_advance; with
 * defer_advance = 0 the launch advances the device-resident nonce itself).  For t = 1 and more than three words per
 * thread of the (capped) grid, a thread's keystream continues from word to word: the coefficient a given position
 * receives then depends on n and on the number of compute units -- as with every device-CSPRNG entry point, the
 * randomness is not part of the parity contract (the opened values are).  Scalars: canonical 2-limb host scalars.  */
int ffgpu_gf256_sbox_layer(ffgpu_ctx* ctx, const uint64_t* host_matrix, const uint64_t* host_bias, const uint64_t* host_lambda,
                           const uint64_t* host_mu, int t, int m, const void* x, size_t x_stride, const void* rbits,
                           size_t rbits_stride, void* out, size_t out_stride, size_t n, const uint8_t* host_key32, uint64_t nonce,
                           int rounds, void* dev_state, int defer_advance, void* stream);

/* out[h] = A * bits(in[h]^254) + B packed back to a byte, with the 8x8 GF(2)
 * matrix given as 8 row bytes (bit c of host_rows8[r] = A[r][c]) and B as a byte.
 * replaces: demos/np_aes.py:37-43 sbox() evaluated on public values.           */
int ffgpu_gf256_sbox(ffgpu_ctx* ctx, const void* in, const uint8_t* host_rows8, uint8_t b,
                     void* out, size_t n, void* stream);

/* ---- timing helper ------------------------------------------------------ */
/* Runs `reps` back-to-back launches of ffgpu_mul on `stream` bracketed by HIP
 * events on that stream and returns the mean milliseconds per launch.
 * (bench-side convenience; equivalent entry points exist for split/recombine) */
int ffgpu_time_mul(ffgpu_ctx* ctx, const void* a, const void* b, void* out, size_t n,
                   int reps, void* stream, float* ms_per_launch);
int ffgpu_time_split(ffgpu_ctx* ctx, const void* secrets, const void* coeffs, size_t coeff_stride,
                     int t, int m, void* shares, size_t share_stride, size_t n,
                     int reps, void* stream, float* ms_per_launch);
int ffgpu_time_recombine(ffgpu_ctx* ctx, const void* const* host_rows, const uint64_t* host_lambda,
                         int k, int w, void* out, size_t out_stride, size_t n,
                         int reps, void* stream, float* ms_per_launch);
/* device-to-device copy of `bytes` (multiple of 16, 16-byte aligned) with the library's own
 * streaming kernel: the achievable-bandwidth yardstick reported next to the 8 TB/s nominal peak. */
int ffgpu_copy(ffgpu_ctx* ctx, const void* src, void* dst, size_t bytes, void* stream);
/* VALU issue-rate yardstick (the compute-side counterpart of ffgpu_time_copy): what the chip sustains for one instruction
 * kind -- op 0: v_bitop3_b32, 1: v_add_u32, 2: v_mad_u64_u32, 3: v_xor_b32, 4: v_perm_b32, 5: v_lshrrev_b32, 6: v_and_or_b32,
 * 7: v_add3_u32, 8: v_mul_lo_u32, 9: v_alignbit_b32 (rotate), 10: v_lshl_or_b32, 11: v_alignbyte_b32, 12: v_lshrrev_b64, 13: v_lshl_add_u64 -- with `waves_per_simd` waves on every SIMD and `iters` x 128
 * instructions in 8 independent dependent-chains per wave.  out3[0] = lane-operations per second (events around the launch),
 * out3[1] = shader clock in MHz under that load (s_memtime against the 100 MHz counter), out3[2] = shader cycles per wave
 * instruction and SIMD.  scratch32: 32 bytes of device memory.  bench.py prices its VALU-bound rows with it.
 * SYNCHRONISES `stream` (event wait + a blocking read-back of the cycle counts): not legal during stream capture, unlike
 * the computing entry points.
 * replaces: nothing in the reference (measurement aid, like the ffgpu_time_* entry points).                            */
int ffgpu_valu_probe(ffgpu_ctx* ctx, int op, int iters, int waves_per_simd, void* scratch32, double* out3, void* stream);
int ffgpu_time_copy(ffgpu_ctx* ctx, const void* src, void* dst, size_t bytes,
                    int reps, void* stream, float* ms_per_launch);

/* ---- device-side wire: share rows between CO-LOCATED party processes -------------------------------------
 * The reference marshals every share row with pickle.dumps and moves the bytes through its asyncio TCP mesh
 * (runtime.py:484-495, 655-661, 571-577; asyncoro.py:54-106): at n = 10^7 that is 80 MB per row and message, 0.5-0.7 s
 * per gate among three local parties against 0.4 ms of kernels.  When the parties are processes on one node (one per
 * GPU, or several on one GPU) the row need not leave the device: the exporting party sends a 64-byte interprocess
 * handle + offset instead of the limb bytes, the receiving party opens it and copies device-to-device (over xGMI
 * between GPUs).  The host side (mpyc_amd/ipcwire.py) keeps the exported buffer alive until every receiver has
 * acknowledged its copy.
 *   export: synchronises `stream` (the row is complete), returns the handle of the ALLOCATION that contains ptr and
 *           ptr's offset in it (hipMemGetAddressRange + hipIpcGetMemHandle);
 *   open:   maps the exporter's allocation into this process (hipIpcOpenMemHandle; not valid in the exporting
 *           process itself -- the host side short-circuits that case); close unmaps it;
 *   read:   copies `bytes` from base + offset into dst and synchronises `stream`.
 *   canary: export also returns the first and last 16 bytes of the row (canary32, may be NULL); read, given them
 *           (expect_canary32, may be NULL), compares them with what the mapping shows BEFORE copying and returns
 *           FFGPU_ESTALE on a mismatch -- the guard for mappings a receiver keeps cached across messages, should an
 *           exporter ever free an allocation and obtain the same handle bytes for a new one.
 * replaces: pickle.dumps(row) / pickle.loads(bytes) of the np path for co-located parties.                      */
#define FFGPU_IPC_HANDLE_BYTES 64
int ffgpu_ipc_export(ffgpu_ctx* ctx, const void* ptr, size_t bytes, unsigned char* handle, unsigned long long* offset,
                     unsigned char* canary32, void* stream);
int ffgpu_ipc_open(ffgpu_ctx* ctx, const unsigned char* handle, void** base);
int ffgpu_ipc_read(ffgpu_ctx* ctx, const void* base, unsigned long long offset, void* dst, size_t bytes,
                   const unsigned char* expect_canary32, void* stream);
/* read + ffgpu_reduce in ONE pass: n field elements are read through the mapping, reduced to canonical form (a peer's data
 * must be canonical before any kernel computes on it, as `field.array(unmarshal(r))` guarantees at runtime.py:508, 677)
 * and written to dst; synchronises `stream`. */
int ffgpu_ipc_read_reduced(ffgpu_ctx* ctx, const void* base, unsigned long long offset, void* dst, size_t n,
                           const unsigned char* expect_canary32, void* stream);
int ffgpu_ipc_close(ffgpu_ctx* ctx, void* base);

#ifdef __cplusplus
}
#endif
#endif /* FFGPU_H */
