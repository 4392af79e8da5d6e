// mfma_probe.hip -- find the operand layout of v_mfma_i32_32x32x32_i8 on gfx950 empirically:
// D = A(32x32) * B(32x32) with random int8 operands, candidate lane->(row, k) mappings, compared with the CPU.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

__global__ void k(const int8_t* A, const int8_t* B, int* D, int variant) {
    int l = threadIdx.x;
    int8_t a[16], b[16];
    for (int e = 0; e < 16; ++e) {
        int kk;
        if (variant == 0) kk = 16 * (l >> 5) + e;                       // two contiguous halves of 16
        else kk = (e < 8) ? 8 * (l >> 5) + e : 16 + 8 * (l >> 5) + (e - 8);   // interleaved groups of 8
        a[e] = A[(l & 31) * 32 + kk];      // A[i][k], i = l & 31
        b[e] = B[kk * 32 + (l & 31)];      // B[k][j], j = l & 31
    }
    v4i av, bv;
    __builtin_memcpy(&av, a, 16);
    __builtin_memcpy(&bv, b, 16);
    v16i c = {0};
    c = __builtin_amdgcn_mfma_i32_32x32x32_i8(av, bv, c, 0, 0, 0);
    for (int r = 0; r < 16; ++r) {
        int col = l & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
        D[row * 32 + col] = c[r];
    }
}
int main() {
    int8_t hA[1024], hB[1024];
    srand(1);
    for (int i = 0; i < 1024; ++i) { hA[i] = (int8_t)(rand() % 255 - 127); hB[i] = (int8_t)(rand() % 255 - 127); }
    int ref[1024];
    for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) { int s = 0; for (int kk = 0; kk < 32; ++kk) s += hA[i * 32 + kk] * hB[kk * 32 + j]; ref[i * 32 + j] = s; }
    int8_t *dA, *dB; int* dD;
    hipMalloc(&dA, 1024); hipMalloc(&dB, 1024); hipMalloc(&dD, 4096);
    hipMemcpy(dA, hA, 1024, hipMemcpyHostToDevice); hipMemcpy(dB, hB, 1024, hipMemcpyHostToDevice);
    for (int v = 0; v < 2; ++v) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dD, v);
        int hD[1024];
        hipMemcpy(hD, dD, 4096, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int i = 0; i < 1024; ++i) bad += hD[i] != ref[i];
        printf("variant %d: %d mismatches\n", v, bad);
    }
    return 0;
}
