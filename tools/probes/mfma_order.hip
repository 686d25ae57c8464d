// Probe: in which order does v_mfma_f32_16x16x4_f32 accumulate its four k?  (The arithmetic contract of the conv
// kernels is an fmaf chain over ascending k; v_mfma_f32_32x32x2_f32 is known to be exactly that for its two k.)
// Build: hipcc --offload-arch=gfx950 -O2 -ffp-contract=off mfma_order.hip -o mfma_order ; run on an MI355X.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ void k16(const float *A, const float *B, const float *C, float *D) // A[16][4], B[4][16], C/D[16][16]
{
    const int l = threadIdx.x;
    f32x4 c;
    for (int r = 0; r < 4; ++r) c[r] = C[(4 * (l / 16) + r) * 16 + l % 16];
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(A[(l % 16) * 4 + l / 16], B[(l / 16) * 16 + l % 16], c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) D[(4 * (l / 16) + r) * 16 + l % 16] = c[r];
}
__global__ void k32(const float *A, const float *B, const float *C, float *D) // A[32][2], B[2][32], C/D[32][32]
{
    const int l = threadIdx.x;
    f32x16 c;
    for (int r = 0; r < 16; ++r) c[r] = C[((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 32 + (l & 31)];
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(A[(l & 31) * 2 + (l >> 5)], B[(l >> 5) * 32 + (l & 31)], c, 0, 0, 0);
    for (int r = 0; r < 16; ++r) D[((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = c[r];
}

static float rnd()
{
    const float m = (float)rand() / RAND_MAX * 2.0f - 1.0f;
    return ldexpf(m, rand() % 24 - 12); // wide exponent range: the order of the roundings shows
}

int main()
{
    srand(1);
    float *dA, *dB, *dC, *dD;
    hipMalloc(&dA, 4096); hipMalloc(&dB, 4096); hipMalloc(&dC, 4096); hipMalloc(&dD, 4096);
    const int perms[6][4] = {{0, 1, 2, 3}, {3, 2, 1, 0}, {0, 2, 1, 3}, {1, 0, 3, 2}, {2, 3, 0, 1}, {0, 1, 3, 2}};
    long bad16[8] = {0}, bad32[2] = {0}, total16 = 0, total32 = 0;
    for (int trial = 0; trial < 200; ++trial) {
        float A[64], B[64], C[256], D[256];
        for (float &v : A) v = rnd();
        for (float &v : B) v = rnd();
        for (float &v : C) v = rnd();
        hipMemcpy(dA, A, sizeof(A), hipMemcpyHostToDevice); hipMemcpy(dB, B, sizeof(B), hipMemcpyHostToDevice);
        hipMemcpy(dC, C, sizeof(C), hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k16, dim3(1), dim3(64), 0, 0, dA, dB, dC, dD);
        hipMemcpy(D, dD, sizeof(D), hipMemcpyDeviceToHost);
        for (int i = 0; i < 16; ++i)
            for (int j = 0; j < 16; ++j) {
                ++total16;
                for (int p = 0; p < 6; ++p) { // sequential fmaf chains in a few k orders
                    float acc = C[i * 16 + j];
                    for (int q = 0; q < 4; ++q) acc = fmaf(A[i * 4 + perms[p][q]], B[perms[p][q] * 16 + j], acc);
                    if (memcmp(&acc, &D[i * 16 + j], 4)) ++bad16[p];
                }
                { // pairwise: (k0 + k1) then (k2 + k3) in separate chains, summed
                    const float t = fmaf(A[i * 4 + 1], B[16 + j], A[i * 4] * B[j]);
                    const float u = fmaf(A[i * 4 + 3], B[48 + j], A[i * 4 + 2] * B[32 + j]);
                    const float acc = C[i * 16 + j] + (t + u);
                    if (memcmp(&acc, &D[i * 16 + j], 4)) ++bad16[6];
                }
                { // exact dot product rounded once
                    double acc = C[i * 16 + j];
                    for (int q = 0; q < 4; ++q) acc += (double)A[i * 4 + q] * B[q * 16 + j];
                    const float f = (float)acc;
                    if (memcmp(&f, &D[i * 16 + j], 4)) ++bad16[7];
                }
            }
    }
    for (int trial = 0; trial < 50; ++trial) {
        float A[64], B[64], C[1024], D[1024];
        for (float &v : A) v = rnd();
        for (float &v : B) v = rnd();
        for (float &v : C) v = rnd();
        hipMemcpy(dA, A, sizeof(A), hipMemcpyHostToDevice); hipMemcpy(dB, B, sizeof(B), hipMemcpyHostToDevice);
        hipMemcpy(dC, C, sizeof(C), hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k32, dim3(1), dim3(64), 0, 0, dA, dB, dC, dD);
        hipMemcpy(D, dD, sizeof(D), hipMemcpyDeviceToHost);
        for (int i = 0; i < 32; ++i)
            for (int j = 0; j < 32; ++j) {
                ++total32;
                for (int p = 0; p < 2; ++p) {
                    float acc = C[i * 32 + j];
                    for (int q = 0; q < 2; ++q) { const int k = p ? 1 - q : q; acc = fmaf(A[i * 2 + k], B[k * 32 + j], acc); }
                    if (memcmp(&acc, &D[i * 32 + j], 4)) ++bad32[p];
                }
            }
    }
    printf("16x16x4 f32: %ld outputs; mismatches vs fmaf chain k=0123: %ld, 3210: %ld, 0213: %ld, 1032: %ld, 2301: %ld, 0132: %ld; pairwise: %ld; exact-dot: %ld\n",
           total16, bad16[0], bad16[1], bad16[2], bad16[3], bad16[4], bad16[5], bad16[6], bad16[7]);
    printf("32x32x2 f32: %ld outputs; mismatches vs fmaf chain k=01: %ld, 10: %ld\n", total32, bad32[0], bad32[1]);
    return 0;
}
