// Probe: issue rate of f32 FMA forms on one SIMD (gfx950): v_fmac_f32 / v_pk_fma_f32, second operand from an SGPR or a VGPR,
// 1..4 waves per SIMD.  Prints cycles per wave-instruction (s_memtime is 100 MHz: converted with the measured shader clock is
// not needed -- the ratio between the forms is what matters; absolute numbers assume --clock GHz).
//   hipcc --offload-arch=gfx950 -O3 tools/probes/valu_fma_rate.hip -o /tmp/valu_fma_rate && /tmp/valu_fma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int FORM>
__global__ __launch_bounds__(256) void k(float *out, const float *w, int iters, unsigned long long *cyc)
{
    float acc[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) acc[i] = (float)threadIdx.x + i;
    float x = out[threadIdx.x], y = out[threadIdx.x + 256];
    const float s0 = w[0], s1 = w[1]; // uniform -> SGPRs
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
#pragma unroll
            for (int i = 0; i < 32; i += 2) {
                if (FORM == 0) { // v_fmac_f32, SGPR weight
                    asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(acc[i]) : "s"(s0), "v"(x));
                    asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(acc[i + 1]) : "s"(s1), "v"(x));
                } else if (FORM == 1) { // v_fmac_f32, VGPR weight
                    asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(acc[i]) : "v"(y), "v"(x));
                    asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(acc[i + 1]) : "v"(y), "v"(x));
                } else if (FORM == 2) { // v_pk_fma_f32, SGPR pair weight, x broadcast
                    typedef float f2 __attribute__((ext_vector_type(2)));
                    f2 a2 = {acc[i], acc[i + 1]};
                    f2 xx = {x, y};
                    asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(a2) : "s"(*(const f2 *)w), "v"(xx));
                    acc[i] = a2.x; acc[i + 1] = a2.y;
                } else { // v_pk_fma_f32, all VGPR
                    typedef float f2 __attribute__((ext_vector_type(2)));
                    f2 a2 = {acc[i], acc[i + 1]};
                    f2 xx = {x, y};
                    asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a2) : "v"(xx), "v"(xx));
                    acc[i] = a2.x; acc[i + 1] = a2.y;
                }
            }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float sum = 0;
#pragma unroll
    for (int i = 0; i < 32; ++i) sum += acc[i];
    out[blockIdx.x * 256 + threadIdx.x] = sum;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

int main()
{
    float *out, *w; unsigned long long *cyc;
    hipMalloc(&out, 4096 * 256 * 4); hipMalloc(&w, 64); hipMalloc(&cyc, 4096 * 8);
    hipMemset(out, 0, 4096 * 256 * 4); hipMemset(w, 0, 64);
    const int iters = 200;
    const char *names[4] = {"v_fmac_f32 sgpr", "v_fmac_f32 vgpr", "v_pk_fma_f32 sgpr", "v_pk_fma_f32 vgpr"};
    for (int form = 0; form < 4; ++form)
        for (int bpc = 1; bpc <= 4; ++bpc) { // blocks of 256 threads per CU = waves per SIMD
            const int grid = 256 * bpc;
            hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
            auto launch = [&]() {
                if (form == 0) hipLaunchKernelGGL(k<0>, dim3(grid), dim3(256), 0, 0, out, w, iters, cyc);
                else if (form == 1) hipLaunchKernelGGL(k<1>, dim3(grid), dim3(256), 0, 0, out, w, iters, cyc);
                else if (form == 2) hipLaunchKernelGGL(k<2>, dim3(grid), dim3(256), 0, 0, out, w, iters, cyc);
                else hipLaunchKernelGGL(k<3>, dim3(grid), dim3(256), 0, 0, out, w, iters, cyc);
            };
            launch(); hipDeviceSynchronize();
            hipEventRecord(a); launch(); hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            const double instr = (double)iters * 8 * (form < 2 ? 32 : 16); // wave-instructions per wave
            // per SIMD: bpc waves, each `instr` instructions, in ms -> ns per instruction slot on the SIMD
            const double ns_per = ms * 1e6 / (instr * bpc);
            const double fma_per_s = (double)grid * 256 * iters * 8 * 32 / (ms * 1e-3);
            printf("%-20s %d wave(s)/SIMD: %.3f ms, %.2f ns per wave-instruction and SIMD (= %.1f cycles at 2.4 GHz), %.1f TFLOP/s chip-wide\n",
                   names[form], bpc, ms, ns_per, ns_per * 2.4, 2 * fma_per_s / 1e12);
        }
    return 0;
}
