// Probe: how many 256-thread blocks with X bytes of dynamic LDS are really resident per CU?
// Each block stamps its start (100 MHz clock) and spins ~100 us; blocks that start late were not resident.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
__global__ __launch_bounds__(256) void probe(unsigned long long *start, int spin_ticks)
{
    extern __shared__ float lds[];
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    lds[threadIdx.x] = (float)t0;
    while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)spin_ticks) __builtin_amdgcn_s_sleep(8);
    if (threadIdx.x == 0) start[blockIdx.x] = t0 + (lds[1] == 12345.0f);
}
int main()
{
    int ncu = 0;
    hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0);
    unsigned long long *d;
    hipMalloc(&d, 8 * 4096);
    for (int q : {96, 128, 144, 160, 176, 192, 200, 204, 208, 209, 210, 211, 212, 213, 214, 216, 224, 240, 256, 288, 320, 384, 448}) { // quarter KB
        const size_t bytes = (size_t)q * 256;
        const double kb = q / 4.0;
        hipFuncSetAttribute((const void *)probe, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        int occ = 0;
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, probe, 256, bytes);
        const int nblk = ncu * 8;
        probe<<<nblk, 256, bytes>>>(d, 10000);
        hipDeviceSynchronize();
        std::vector<unsigned long long> h(nblk);
        hipMemcpy(h.data(), d, 8 * nblk, hipMemcpyDeviceToHost);
        const unsigned long long t0 = *std::min_element(h.begin(), h.end());
        int first = 0;
        for (auto t : h) first += (t - t0 < 2000); // started within 20 us of the first block
        printf("%6.2f KB: occupancy API %d blocks/CU, measured %.2f blocks/CU resident at once (%d CUs)\n", kb, occ, (double)first / ncu, ncu);
    }
    return 0;
}
