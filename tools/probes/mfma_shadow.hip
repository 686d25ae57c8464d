// Probe: how many independent VALU / LDS instructions fit "under" a v_mfma_f32_32x32x2_f32 (16 passes = 64 cycles)
// when ONE wave per SIMD issues them in order?  Build: hipcc --offload-arch=gfx950 -O3 mfma_shadow.hip -o mfma_shadow
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int K, int MODE, int NT = 256> // MODE 4: K global_load_dword, 5: K SALU, 6: K ds_write_b64; MODE 0: K v_add after every MFMA; 1: 10*K v_add after every 10th MFMA; 2: K ds_read_b64 after every MFMA; 3: K v_mul_lo_u32
__global__ __launch_bounds__(NT) void probe(float *out, unsigned long long *cyc, int iters, const float *src)
{
    __shared__ float2 lds[1024];
    const int lane = threadIdx.x & 63;
    lds[threadIdx.x] = make_float2(threadIdx.x, 1.0f);
    float g[4] = {0, 0, 0, 0};
    int sacc = iters;
    __syncthreads();
    f32x16 acc[10];
    for (int i = 0; i < 10; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;
    float a = lane * 0.001f, b = 1.0f + lane * 0.002f;
    unsigned x[16];
    for (int i = 0; i < 16; ++i) x[i] = lane + i;
    float2 ld[4] = {{0, 0}, {0, 0}, {0, 0}, {0, 0}};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 10; ++i) {
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
            if (MODE == 0) {
#pragma unroll
                for (int k = 0; k < K; ++k) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x[k % 16]) : "v"(lane));
            } else if (MODE == 3) {
#pragma unroll
                for (int k = 0; k < K; ++k) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(x[k % 16]) : "v"(lane));
            } else if (MODE == 2) {
#pragma unroll
                for (int k = 0; k < K; ++k) ld[k % 4] = lds[(lane + x[k % 16]) & 1023];
            } else if (MODE == 4) {
#pragma unroll
                for (int k = 0; k < K; ++k) g[k % 4] = src[threadIdx.x + 64 * k];
            } else if (MODE == 5) {
#pragma unroll
                for (int k = 0; k < K; ++k) asm volatile("s_mul_i32 %0, %0, 3" : "+s"(sacc));
            } else if (MODE == 6) {
#pragma unroll
                for (int k = 0; k < K; ++k) lds[(threadIdx.x + 256 * k) & 1023] = make_float2(a, b);
            } else if (MODE == 1 && i == 9) {
#pragma unroll
                for (int k = 0; k < 10 * K; ++k) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x[k % 16]) : "v"(lane));
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < 10; ++i)
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    for (int i = 0; i < 16; ++i) s += x[i];
    for (int i = 0; i < 4; ++i) s += ld[i].x + ld[i].y + g[i];
    s += sacc;
    out[blockIdx.x * NT + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int K, int MODE, int NT = 256> void run(const char *name, float *out, unsigned long long *cyc)
{
    const int iters = 2000;
    static float *src = nullptr; if (!src) { hipMalloc(&src, 1 << 20); hipMemset(src, 0, 1 << 20); }
    probe<K, MODE, NT><<<1, NT>>>(out, cyc, iters, src);
    probe<K, MODE, NT><<<1, NT>>>(out, cyc, iters, src);
    hipDeviceSynchronize();
    unsigned long long c;
    hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-28s K=%2d waves/SIMD=%d : %.1f cycles per MFMA of one wave\n", name, K, NT / 256, (double)c / (iters * 10.0));
}

int main()
{
    float *out; unsigned long long *cyc;
    hipMalloc(&out, 512 * 4 * 4); hipMalloc(&cyc, 64);
    run<0, 0>("mfma only", out, cyc);
    run<1, 0>("v_add after each", out, cyc);  run<2, 0>("v_add after each", out, cyc);  run<4, 0>("v_add after each", out, cyc);
    run<8, 0>("v_add after each", out, cyc);  run<12, 0>("v_add after each", out, cyc); run<16, 0>("v_add after each", out, cyc);
    run<1, 1>("10K v_add after 10 mfma", out, cyc); run<2, 1>("10K v_add after 10 mfma", out, cyc); run<4, 1>("10K v_add after 10 mfma", out, cyc);
    run<1, 2>("ds_read_b64 after each", out, cyc); run<2, 2>("ds_read_b64 after each", out, cyc); run<4, 2>("ds_read_b64 after each", out, cyc);
    run<1, 3>("v_mul_lo_u32 after each", out, cyc); run<2, 3>("v_mul_lo_u32 after each", out, cyc); run<4, 3>("v_mul_lo_u32 after each", out, cyc);
    run<1, 4>("global_load after each", out, cyc); run<2, 4>("global_load after each", out, cyc); run<4, 4>("global_load after each", out, cyc);
    run<1, 5>("s_mul after each", out, cyc); run<4, 5>("s_mul after each", out, cyc);
    run<1, 6>("ds_write_b64 after each", out, cyc); run<2, 6>("ds_write_b64 after each", out, cyc);
    run<0, 0, 512>("mfma only", out, cyc);
    run<4, 0, 512>("v_add after each", out, cyc); run<8, 0, 512>("v_add after each", out, cyc); run<16, 0, 512>("v_add after each", out, cyc);
    return 0;
}
