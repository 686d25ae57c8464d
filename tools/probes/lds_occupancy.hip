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
    for (int kb : {24, 32, 36, 40, 44, 48, 50, 51, 52, 53, 54, 56, 60, 64, 72, 80, 96, 112}) {
        const size_t bytes = (size_t)kb * 1024;
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
        printf("%3d KB: occupancy API %d blocks/CU, measured %.2f blocks/CU resident at once (%d CUs)\n", kb, occ, (double)first / ncu, ncu);
    }
    return 0;
}
