// tools/shake_dev.hip -- MEASUREMENT PROBE (not part of the product): SHAKE128 on the device, to put a number on
// "PRSS XOF on the GPU" (thresha.py:238-266: one SHAKE128 stream per subset key, n * byte_length output bytes each).
//
// A sponge is sequential per stream (every 168-byte block is one Keccak-f[1600] of the previous state), so the only
// parallelism PRSS offers is ACROSS the subset keys of a call: C(m-1, t) streams for party i (20 at m = 7, t = 3).
// Mapping measured here: ONE LANE PER STREAM, the 25 64-bit lanes of the state in 50 VGPRs, rounds fully unrolled
// (rotations are v_alignbit pairs, chi is xor + v_bfi) -- the mapping with the fewest instructions per permutation and
// no cross-lane traffic.  It prints bytes/s per stream and in total for S streams, and digests the host checks against
// hashlib (tools/shake_dev_check.py).
//
// build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 -o /tmp/shake_dev tools/shake_dev.hip && /tmp/shake_dev
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ uint64_t rotl(uint64_t v, int r) { return (v << r) | (v >> (64 - r)); }

__constant__ uint64_t RC[24] = {
    0x0000000000000001ull, 0x0000000000008082ull, 0x800000000000808aull, 0x8000000080008000ull, 0x000000000000808bull,
    0x0000000080000001ull, 0x8000000080008081ull, 0x8000000000008009ull, 0x000000000000008aull, 0x0000000000000088ull,
    0x0000000080008009ull, 0x000000008000000aull, 0x000000008000808bull, 0x800000000000008bull, 0x8000000000008089ull,
    0x8000000000008003ull, 0x8000000000008002ull, 0x8000000000000080ull, 0x000000000000800aull, 0x800000008000000aull,
    0x8000000080008081ull, 0x8000000000008080ull, 0x0000000080000001ull, 0x8000000080008008ull};

// Keccak-f[1600] on 25 named registers (FIPS 202 section 3.2; lane (x, y) is a[x + 5 y])
__device__ __forceinline__ void keccak_f(uint64_t (&a)[25]) {
#pragma unroll 1
    for (int rnd = 0; rnd < 24; ++rnd) {
        uint64_t c[5], d[5], b[25];
#pragma unroll
        for (int x = 0; x < 5; ++x) c[x] = a[x] ^ a[x + 5] ^ a[x + 10] ^ a[x + 15] ^ a[x + 20];
#pragma unroll
        for (int x = 0; x < 5; ++x) d[x] = c[(x + 4) % 5] ^ rotl(c[(x + 1) % 5], 1);
#pragma unroll
        for (int i = 0; i < 25; ++i) a[i] ^= d[i % 5];
        // rho + pi: b[y, 2x + 3y] = rotl(a[x, y], r[x, y])
        b[0] = a[0];
        b[10] = rotl(a[1], 1);   b[20] = rotl(a[2], 62);  b[5] = rotl(a[3], 28);   b[15] = rotl(a[4], 27);
        b[16] = rotl(a[5], 36);  b[1] = rotl(a[6], 44);   b[11] = rotl(a[7], 6);   b[21] = rotl(a[8], 55);
        b[6] = rotl(a[9], 20);   b[7] = rotl(a[10], 3);   b[17] = rotl(a[11], 10); b[2] = rotl(a[12], 43);
        b[12] = rotl(a[13], 25); b[22] = rotl(a[14], 39); b[23] = rotl(a[15], 41); b[8] = rotl(a[16], 45);
        b[18] = rotl(a[17], 15); b[3] = rotl(a[18], 21);  b[13] = rotl(a[19], 8);  b[14] = rotl(a[20], 18);
        b[24] = rotl(a[21], 2);  b[9] = rotl(a[22], 61);  b[19] = rotl(a[23], 56); b[4] = rotl(a[24], 14);
#pragma unroll
        for (int y = 0; y < 25; y += 5)
#pragma unroll
            for (int x = 0; x < 5; ++x) a[y + x] = b[y + x] ^ (~b[y + (x + 1) % 5] & b[y + (x + 2) % 5]);
        a[0] ^= RC[rnd];
    }
}

// stream s: message = 16 key bytes ((7 s + j) & 0xff, j = 0..15) followed by "uci" -- one absorbed block; then `blocks`
// squeezed blocks of 168 bytes are written to out + s * blocks * 168 (21 words per block, stream-major)
__global__ __launch_bounds__(64) void k_shake(uint64_t* __restrict__ out, int streams, int blocks) {
    const int s = blockIdx.x * 64 + threadIdx.x;
    if (s >= streams) return;
    uint64_t a[25];
#pragma unroll
    for (int i = 0; i < 25; ++i) a[i] = 0;
    unsigned char msg[24];
    for (int j = 0; j < 16; ++j) msg[j] = (unsigned char)((7 * s + j) & 0xff);
    msg[16] = 'u'; msg[17] = 'c'; msg[18] = 'i';
    msg[19] = 0x1f;                                   // SHAKE domain separation + first padding bit
    for (int j = 20; j < 24; ++j) msg[j] = 0;
    for (int w = 0; w < 3; ++w) {
        uint64_t v = 0;
        for (int j = 0; j < 8; ++j) v |= (uint64_t)msg[8 * w + j] << (8 * j);
        a[w] = v;
    }
    a[20] ^= 0x8000000000000000ull;                   // last padding bit of the 168-byte rate
    uint64_t* o = out + (size_t)s * blocks * 21;
    for (int b = 0; b < blocks; ++b) {
        keccak_f(a);
#pragma unroll
        for (int i = 0; i < 21; ++i) o[(size_t)b * 21 + i] = a[i];
    }
}

int main(int argc, char** argv) {
    const size_t budget = argc > 1 ? (size_t)atoll(argv[1]) : ((size_t)256 << 20);   // output bytes per measurement
    const int counts[] = {1, 20, 64, 256, 1024, 4096, 16384, 65536, 262144};
    uint64_t* out;
    CHECK(hipMalloc(&out, budget + 4096));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    // digests for the host check: streams 0, 19 of the 20-stream case, 1000 blocks each
    {
        const int S = 20, B = 1000;
        k_shake<<<(S + 63) / 64, 64>>>(out, S, B);
        CHECK(hipDeviceSynchronize());
        std::vector<uint64_t> h((size_t)S * B * 21);
        CHECK(hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost));
        for (int s : {0, 19}) {
            uint64_t x = 0;
            for (size_t i = 0; i < (size_t)B * 21; ++i) x ^= h[(size_t)s * B * 21 + i] * (2 * i + 1);
            printf("CHECK stream %d first16 ", s);
            const unsigned char* pb = (const unsigned char*)&h[(size_t)s * B * 21];
            for (int j = 0; j < 16; ++j) printf("%02x", pb[j]);
            printf(" fold %016llx\n", (unsigned long long)x);
        }
    }
    for (int S : counts) {
        size_t per = budget / S / 168;
        if (per > 20000) per = 20000;                 // <= 3.4 MB per stream per measurement: bounded run time
        if (per < 1) break;
        const int B = (int)per;
        k_shake<<<(S + 63) / 64, 64>>>(out, S, B < 50 ? B : 50);      // warm
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0));
        k_shake<<<(S + 63) / 64, 64>>>(out, S, B);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        const double bytes = (double)S * B * 168;
        printf("streams %7d blocks/stream %6d  %9.3f ms  per stream %8.2f MB/s  total %9.3f GB/s  (%.2f us per Keccak-f per lane)\n", S, B,
               ms, (double)B * 168 / ms / 1e3, bytes / ms / 1e6, ms * 1e3 / B);
    }
    return 0;
}
