// Does a wave's MFMA stream take VALU issue slots from ANOTHER wave on the same SIMD?  (round 4: the converter on the
// matrix cores, profiles/ab_r04.txt section 10)
// One workgroup of 8 waves per CU: waves 0-3 (one per SIMD) run a dependent-free VALU loop and time it; waves 4-7 (their SIMD
// partners) run, by mode: 0 nothing, 1 v_mfma_f32_16x16x4_f32 (fp32 in), 2 v_mfma_f32_32x32x16_f16, 3 the same VALU loop.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_coissue mfma_coissue.hip && ./mfma_coissue
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

__global__ __launch_bounds__(512) void k(int mode, int iters, unsigned long long *cycles, float *sink)
{
    const int wave = threadIdx.x >> 6;
    float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    const float a = 1.0001f, b = 0.5f;
    __syncthreads();
    if (wave < 4 || mode == 3) {
        unsigned long long t0 = __builtin_readcyclecounter();
        for (int i = 0; i < iters; i++) {
            x0 = __builtin_fmaf(x0, a, b); x1 = __builtin_fmaf(x1, a, b); x2 = __builtin_fmaf(x2, a, b); x3 = __builtin_fmaf(x3, a, b);
            x4 = __builtin_fmaf(x4, a, b); x5 = __builtin_fmaf(x5, a, b); x6 = __builtin_fmaf(x6, a, b); x7 = __builtin_fmaf(x7, a, b);
        }
        unsigned long long t1 = __builtin_readcyclecounter();
        if ((threadIdx.x & 63) == 0 && wave < 4) cycles[blockIdx.x * 4 + wave] = t1 - t0;
    } else if (mode == 1) {
        f32x4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
        for (int i = 0; i < iters / 8; i++) {          // 16 x 32 cycles of the pipe per iteration
            for (int j = 0; j < 8; j++) {
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x0, x1, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x2, x3, acc1, 0, 0, 0);
            }
        }
        x0 = acc0[0] + acc1[1];
    } else if (mode == 2) {
        f32x16 acc = {0};
        f16x8 ha = {(_Float16)x0, (_Float16)x1, 1, 2, 3, 4, 5, 6}, hb = {(_Float16)x2, 2, 3, 4, 5, 6, 7, 8};
        for (int i = 0; i < iters / 8; i++) {
            for (int j = 0; j < 16; j++) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, hb, acc, 0, 0, 0);
        }
        x0 = acc[0] + acc[5];
    }
    sink[blockIdx.x * 512 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}

int main()
{
    const int blocks = 256, iters = 20000;
    unsigned long long *d, h[blocks * 4];
    float *sink;
    hipMalloc(&d, sizeof h);
    hipMalloc(&sink, blocks * 512 * sizeof(float));
    const char *names[] = {"partner idle", "partner v_mfma_f32_16x16x4_f32 (two chains)", "partner v_mfma_f32_32x32x16_f16", "partner the same VALU loop"};
    for (int mode = 0; mode < 4; mode++) {
        for (int rep = 0; rep < 2; rep++) {
            hipLaunchKernelGGL(k, dim3(blocks), dim3(512), 0, 0, mode, iters, d, sink);
            hipDeviceSynchronize();
        }
        hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
        double s = 0;
        for (int i = 0; i < blocks * 4; i++) s += (double)h[i];
        printf("%-48s %.2f cycles per v_fma_f32 of the timed wave\n", names[mode], s / (blocks * 4) / (iters * 8.0));
    }
    return 0;
}
