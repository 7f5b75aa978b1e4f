// Do two workgroups really share a CU at a given LDS size / thread count?  512 workgroups (2 per CU on 256 CUs) that
// each spin for a fixed number of cycles: elapsed ~ 1x the spin when they are co-resident, ~ 2x when the second waits.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#ifndef CLOBBER
#define CLOBBER "v121"
#endif
__global__ __launch_bounds__(384) void spin(unsigned long long cycles, float *sink)
{
    asm volatile("" ::: CLOBBER, "s91");          // the tube kernel's register footprint: 122 VGPRs, 92 SGPRs
    extern __shared__ float lds[];
    lds[threadIdx.x] = (float)threadIdx.x;
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    while (__builtin_readcyclecounter() - t0 < cycles) __builtin_amdgcn_s_sleep(8);
    if (threadIdx.x == 0) sink[blockIdx.x] = lds[threadIdx.x];
}
int main(int argc, char **argv)
{
    int threads = argc > 1 ? atoi(argv[1]) : 384;
    float *sink; hipMalloc(&sink, 4096 * 4);
    hipFuncSetAttribute((const void *)spin, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int kb : {16, 32, 48, 60, 64, 65, 70, 74, 76, 78, 80, 96, 112}) {
        for (int grid : {256, 512}) {
            hipLaunchKernelGGL(spin, dim3(grid), dim3(threads), kb * 1024, 0, 1000ull, sink);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            hipLaunchKernelGGL(spin, dim3(grid), dim3(threads), kb * 1024, 0, 100000ull, sink);   // 100 k cycles of the 100 MHz counter?
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            int occ = 0; hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, spin, threads, kb * 1024);
            printf("lds %3d KB threads %d grid %d: %.3f ms (occupancy api %d)%s", kb, threads, grid, ms, occ, grid == 512 ? "\n" : "   |   ");
        }
    }
    return 0;
}
