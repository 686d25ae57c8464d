// Probe (round 5): what would the conv GEMMs gain -- and lose -- if their f32 operands were split into three bf16 parts and six bf16 products
// accumulated in f32 (hi*hi, hi*mid, mid*hi, hi*lo, lo*hi, mid*mid) instead of one f32 MFMA?
//   1. rate: cycles per MFMA of v_mfma_f32_16x16x4_f32 (what the tile kernels use), v_mfma_f32_16x16x16_bf16 and gfx950's v_mfma_f32_16x16x32_bf16,
//      one wave per SIMD, eight independent accumulators;
//   2. error: D = A B for K = 1152 (conv5 of the e-fcn), A ~ He-init weights, B ~ activated feature values; against float64: the float32 fmaf
//      chain of the arithmetic contract (Numerics) and the six-product bf16 scheme on the MFMA.
// Build: hipcc --offload-arch=gfx950 -O2 -ffp-contract=off -Wno-unused-value mfma_bf16x6.hip -o mfma_bf16x6 ; run on an MI355X.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int KIND> __global__ void k_rate(float *out, unsigned long long *cyc, int iters)
{
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const float a = 1.0f + threadIdx.x * 1e-3f, b = 0.5f;
    s16x4 sa = {0x3f80, 0x3f80, 0x3f80, 0x3f80}, sb = sa;
    bf16x8 ba, bb;
    for (int i = 0; i < 8; ++i) { ba[i] = (__bf16)1.0f; bb[i] = (__bf16)0.5f; }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (KIND == 0) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
            if (KIND == 1) acc[i] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(sa, sb, acc[i], 0, 0, 0);
            if (KIND == 2) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ba, bb, acc[i], 0, 0, 0);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

__device__ __forceinline__ unsigned short bf16_rn(float x)
{
    unsigned u = __float_as_uint(x);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
__device__ __forceinline__ float bf16_f(unsigned short h) { return __uint_as_float((unsigned)h << 16); }

// one wave: D[16][16] = A[16][K] B[K][16]; PRODUCTS = 6: the split scheme, 1: plain bf16 (for scale)
template <int PRODUCTS> __global__ void k_split(const float *A, const float *B, float *D, int K)
{
    const int l = threadIdx.x, i = l % 16, kq = 4 * (l / 16);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int k0 = 0; k0 < K; k0 += 16) {
        s16x4 ah, am, al, bh, bm, bl;
        for (int q = 0; q < 4; ++q) {
            const float a = A[i * K + k0 + kq + q], b = B[(k0 + kq + q) * 16 + i];
            const unsigned short a0 = bf16_rn(a); const float ar = a - bf16_f(a0);
            const unsigned short a1 = bf16_rn(ar); const unsigned short a2 = bf16_rn(ar - bf16_f(a1));
            const unsigned short b0 = bf16_rn(b); const float br = b - bf16_f(b0);
            const unsigned short b1 = bf16_rn(br); const unsigned short b2 = bf16_rn(br - bf16_f(b1));
            ah[q] = (short)a0; am[q] = (short)a1; al[q] = (short)a2; bh[q] = (short)b0; bm[q] = (short)b1; bl[q] = (short)b2;
        }
        if (PRODUCTS == 6) { // small terms first
            acc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(am, bm, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(al, bh, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ah, bl, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(am, bh, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ah, bm, acc, 0, 0, 0);
        }
        acc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ah, bh, acc, 0, 0, 0);
    }
    for (int r = 0; r < 4; ++r) D[(4 * (l / 16) + r) * 16 + l % 16] = acc[r];
}

static double gauss()
{
    const double u = (rand() + 1.0) / (RAND_MAX + 2.0), v = (rand() + 1.0) / (RAND_MAX + 2.0);
    return sqrt(-2.0 * log(u)) * cos(6.283185307179586 * v);
}

int main()
{
    { // ---- rate ----
        float *out; unsigned long long *cyc;
        hipMalloc(&out, 256 * 256 * sizeof(float)); hipMalloc(&cyc, 256 * sizeof(unsigned long long));
        const int iters = 4096;
        const char *names[3] = {"v_mfma_f32_16x16x4_f32  ", "v_mfma_f32_16x16x16_bf16", "v_mfma_f32_16x16x32_bf16"};
        const double flops[3] = {16.0 * 16 * 4 * 2, 16.0 * 16 * 16 * 2, 16.0 * 16 * 32 * 2};
        for (int kind = 0; kind < 3; ++kind) {
            for (int rep = 0; rep < 2; ++rep) {
                if (kind == 0) hipLaunchKernelGGL(k_rate<0>, dim3(256), dim3(256), 0, 0, out, cyc, iters);
                if (kind == 1) hipLaunchKernelGGL(k_rate<1>, dim3(256), dim3(256), 0, 0, out, cyc, iters);
                if (kind == 2) hipLaunchKernelGGL(k_rate<2>, dim3(256), dim3(256), 0, 0, out, cyc, iters);
                hipDeviceSynchronize();
            }
            unsigned long long h[256];
            hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
            double mean = 0;
            for (unsigned long long v : h) mean += (double)v;
            mean /= 256;
            const double per = mean / (iters * 8.0);
            printf("%s  %.1f cycles per MFMA and wave (one wave per SIMD) = %.0f flop per cycle and CU\n", names[kind], per, 4.0 * flops[kind] / per);
        }
    }
    { // ---- error ----
        srand(7);
        const int K = 1152, trials = 200;
        std::vector<float> A(16 * K), B(K * 16), D(256), D1(256);
        float *dA, *dB, *dD;
        hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dD, 1024);
        double e_chain = 0, e_six = 0, e_one = 0, m_chain = 0, m_six = 0, m_one = 0, rms_out = 0;
        long n = 0, bits_equal = 0;
        for (int t = 0; t < trials; ++t) {
            for (float &v : A) v = (float)(gauss() * sqrt(2.0 / K));                          // rule W
            for (float &v : B) { const double g = gauss(); v = (float)(g > 0 ? g : 0.1 * g); } // activated values, alpha = 0.1
            hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
            hipLaunchKernelGGL(k_split<6>, dim3(1), dim3(64), 0, 0, dA, dB, dD, K);
            hipMemcpy(D.data(), dD, 1024, hipMemcpyDeviceToHost);
            hipLaunchKernelGGL(k_split<1>, dim3(1), dim3(64), 0, 0, dA, dB, dD, K);
            hipMemcpy(D1.data(), dD, 1024, hipMemcpyDeviceToHost);
            for (int i = 0; i < 16; ++i)
                for (int j = 0; j < 16; ++j) {
                    double ref = 0; float chain = 0.f;
                    for (int k = 0; k < K; ++k) { ref += (double)A[i * K + k] * B[k * 16 + j]; chain = fmaf(A[i * K + k], B[k * 16 + j], chain); }
                    const double ec = fabs(chain - ref), e6 = fabs(D[i * 16 + j] - ref), e1 = fabs(D1[i * 16 + j] - ref);
                    e_chain += ec * ec; e_six += e6 * e6; e_one += e1 * e1; rms_out += ref * ref;
                    m_chain = fmax(m_chain, ec); m_six = fmax(m_six, e6); m_one = fmax(m_one, e1);
                    bits_equal += memcmp(&chain, &D[i * 16 + j], 4) == 0;
                    ++n;
                }
        }
        printf("K = %d, %ld outputs, rms |output| %.3f: error against float64\n", K, n, sqrt(rms_out / n));
        printf("  float32 fmaf chain (the contract)     rms %.3e  max %.3e\n", sqrt(e_chain / n), m_chain);
        printf("  six bf16 products, f32 accumulation   rms %.3e  max %.3e   (bit-identical to the chain: %ld of %ld)\n", sqrt(e_six / n), m_six, bits_equal, n);
        printf("  one bf16 product (for scale)          rms %.3e  max %.3e\n", sqrt(e_one / n), m_one);
    }
    return 0;
}
