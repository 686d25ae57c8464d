// Probe: how fast does a wave read the SAME 16 bytes into all its lanes (ds_read_b128 with a wave-uniform address), 1..4 waves
// per SIMD, every CU busy?  (For a site-per-lane conv on the vector pipe the weights would arrive this way.)
//   hipcc --offload-arch=gfx950 -O3 tools/probes/lds_broadcast_rate.hip -o /tmp/p && /tmp/p
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
template <int UNIFORM>
__global__ __launch_bounds__(256) void k(float *out, int iters)
{
    __shared__ f4 lds[1024];
    for (int i = threadIdx.x; i < 1024; i += 256) lds[i] = (f4){(float)i, 1.f, 2.f, 3.f};
    __syncthreads();
    f4 acc = {0, 0, 0, 0};
    const int base = UNIFORM ? 0 : threadIdx.x; // uniform: all lanes the same address; else: 16 B per lane, conflict-free
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 32; ++r) {
            f4 v;
            const f4 *p = &lds[(base + r * 8 + (it & 3)) & 1023];
            asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"((unsigned)(size_t)p));
            asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
            acc += v;
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    out[blockIdx.x * 256 + threadIdx.x] = acc.x + acc.y + acc.z + acc.w;
}
int main()
{
    float *out; hipMalloc(&out, 4096 * 256 * 4);
    const int iters = 400;
    for (int uni = 0; uni < 2; ++uni)
        for (int bpc = 1; bpc <= 4; ++bpc) {
            const int grid = 256 * bpc;
            hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
            auto launch = [&]() { if (uni) hipLaunchKernelGGL(k<1>, dim3(grid), dim3(256), 0, 0, out, iters); else hipLaunchKernelGGL(k<0>, dim3(grid), dim3(256), 0, 0, out, iters); };
            launch(); hipDeviceSynchronize();
            hipEventRecord(a); launch(); hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            const double instr_cu = (double)iters * 32 * 4 * bpc; // ds_read_b128 wave-instructions per CU
            printf("%s address, %d wave(s)/SIMD: %.3f ms, %.1f cycles (2.4 GHz) per ds_read_b128 and CU\n", uni ? "uniform" : "per-lane", bpc, ms, ms * 1e6 * 2.4 / instr_cu);
        }
    return 0;
}
