// VALU issue ceiling of a gfx950 SIMD by instruction class and by the number of waves sharing the SIMD.
// One workgroup of 256 * W threads on one CU = W waves on each of the four SIMDs, every wave issuing the same stream
// of N independent instructions of one class (8 rotating destination registers: no dependence stalls) or, for the
// "chain" rows, one dependent chain.  Reported: cycles of the shader clock per instruction PER SIMD
// (= wall cycles / (N * W)): the figure to price a kernel's instruction mix with.  Diagnostic tool only.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>

#define REP 16          /* groups of 8 instructions per loop iteration */
#define ITERS 256

template <int KIND>
__global__ __launch_bounds__(1024) void k(unsigned long long *out, float *sink)
{
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float a0 = lane, a1 = lane + 1, a2 = lane + 2, a3 = lane + 3, a4 = lane + 4, a5 = lane + 5, a6 = lane + 6, a7 = lane + 7;
    float b = 1.0001f, c = 0.5f;
    double d0 = lane, d1 = lane + 1, d2 = lane + 2, d3 = lane + 3, db = 1.0000001;
    typedef float v2f __attribute__((ext_vector_type(2)));
    v2f p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, pb = {b, b};
    int i0 = lane;
    const float sb = __builtin_amdgcn_readfirstlane(1065361605);      // (an SGPR: 1.0001f's bits as int -> float below)
    const unsigned long long smask = __builtin_amdgcn_readfirstlane(0x55555555) * 0x100000001ull;
    const unsigned long long spair = 0x3f8003473f800347ull;
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int r = 0; r < REP; r++) {
            if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
            if (KIND == 1) asm volatile("v_pk_fma_f32 %0, %0, %4, %4\n v_pk_fma_f32 %1, %1, %4, %4\n v_pk_fma_f32 %2, %2, %4, %4\n v_pk_fma_f32 %3, %3, %4, %4\n v_pk_fma_f32 %0, %0, %4, %4\n v_pk_fma_f32 %1, %1, %4, %4\n v_pk_fma_f32 %2, %2, %4, %4\n v_pk_fma_f32 %3, %3, %4, %4" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pb));
            if (KIND == 2) asm volatile("v_fma_f64 %0, %0, %4, %4\n v_fma_f64 %1, %1, %4, %4\n v_fma_f64 %2, %2, %4, %4\n v_fma_f64 %3, %3, %4, %4\n v_fma_f64 %0, %0, %4, %4\n v_fma_f64 %1, %1, %4, %4\n v_fma_f64 %2, %2, %4, %4\n v_fma_f64 %3, %3, %4, %4" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(db));
            if (KIND == 3) asm volatile("v_mov_b32_dpp %0, %1 row_ror:4 row_mask:0xf bank_mask:0x5\n v_mov_b32_dpp %1, %2 row_ror:4 row_mask:0xf bank_mask:0x5\n v_mov_b32_dpp %2, %3 row_ror:4 row_mask:0xf bank_mask:0x5\n v_mov_b32_dpp %3, %4 row_ror:4 row_mask:0xf bank_mask:0x5\n v_mov_b32_dpp %4, %5 row_ror:4 row_mask:0xf bank_mask:0x5\n v_mov_b32_dpp %5, %6 row_ror:4 row_mask:0xf bank_mask:0x5\n v_mov_b32_dpp %6, %7 row_ror:4 row_mask:0xf bank_mask:0x5\n v_mov_b32_dpp %7, %0 row_ror:4 row_mask:0xf bank_mask:0x5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
            if (KIND == 4) asm volatile("v_mov_b32 %0, %8\n v_mov_b32 %1, %8\n v_mov_b32 %2, %8\n v_mov_b32 %3, %8\n v_mov_b32 %4, %8\n v_mov_b32 %5, %8\n v_mov_b32 %6, %8\n v_mov_b32 %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
            if (KIND == 5) asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n v_rcp_f32 %4, %4\n v_rcp_f32 %5, %5\n v_rcp_f32 %6, %6\n v_rcp_f32 %7, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
            if (KIND == 6) asm volatile("v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc");
            if (KIND == 7) asm volatile("v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1" : "+v"(i0) : "v"(lane));
            if (KIND == 8) asm volatile("v_mul_f32 %0, %0, %8\n v_add_f32 %1, %1, %9\n v_mul_f32 %2, %2, %8\n v_add_f32 %3, %3, %9\n v_sub_f32 %4, %4, %9\n v_mul_f32 %5, %5, %8\n v_add_f32 %6, %6, %9\n v_mul_f32 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
            // scalar operands: an SGPR source, a mask in an SGPR pair instead of VCC
            if (KIND == 14) asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "s"(sb), "v"(c));
            if (KIND == 15) asm volatile("v_cndmask_b32_e64 %0, %0, %8, %9\n v_cndmask_b32_e64 %1, %1, %8, %9\n v_cndmask_b32_e64 %2, %2, %8, %9\n v_cndmask_b32_e64 %3, %3, %8, %9\n v_cndmask_b32_e64 %4, %4, %8, %9\n v_cndmask_b32_e64 %5, %5, %8, %9\n v_cndmask_b32_e64 %6, %6, %8, %9\n v_cndmask_b32_e64 %7, %7, %8, %9" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "s"(smask));
            if (KIND == 16) asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "s"(spair), "v"(pb));
            if (KIND == 17) asm volatile("v_cmp_gt_f32 vcc, %0, %8\n v_cndmask_b32 %1, %1, %8, vcc\n v_cmp_gt_f32 vcc, %2, %8\n v_cndmask_b32 %3, %3, %8, vcc\n v_cmp_gt_f32 vcc, %4, %8\n v_cndmask_b32 %5, %5, %8, vcc\n v_cmp_gt_f32 vcc, %6, %8\n v_cndmask_b32 %7, %7, %8, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc");
            // dependent chains (latency): every instruction reads the one before it
            if (KIND == 9) asm volatile("v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2" : "+v"(a0) : "v"(b), "v"(c));
            if (KIND == 10) asm volatile("v_pk_fma_f32 %0, %0, %1, %1\n v_pk_fma_f32 %0, %0, %1, %1\n v_pk_fma_f32 %0, %0, %1, %1\n v_pk_fma_f32 %0, %0, %1, %1\n v_pk_fma_f32 %0, %0, %1, %1\n v_pk_fma_f32 %0, %0, %1, %1\n v_pk_fma_f32 %0, %0, %1, %1\n v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p0) : "v"(pb));
            if (KIND == 11) asm volatile("v_fma_f32 %0, %1, %2, %3\n s_nop 1\n v_mov_b32_dpp %1, %0 row_ror:4 row_mask:0xf bank_mask:0x5\n v_fma_f32 %0, %1, %2, %3\n s_nop 1\n v_mov_b32_dpp %1, %0 row_ror:4 row_mask:0xf bank_mask:0x5\n v_fma_f32 %0, %1, %2, %3\n s_nop 1\n v_mov_b32_dpp %1, %0 row_ror:4 row_mask:0xf bank_mask:0x5\n v_fma_f32 %0, %1, %2, %3\n s_nop 1\n v_mov_b32_dpp %1, %0 row_ror:4 row_mask:0xf bank_mask:0x5" : "+v"(a0), "+v"(a1) : "v"(b), "v"(c));
            if (KIND == 12) asm volatile("v_fma_f64 %0, %0, %1, %1\n v_fma_f64 %0, %0, %1, %1\n v_fma_f64 %0, %0, %1, %1\n v_fma_f64 %0, %0, %1, %1\n v_fma_f64 %0, %0, %1, %1\n v_fma_f64 %0, %0, %1, %1\n v_fma_f64 %0, %0, %1, %1\n v_fma_f64 %0, %0, %1, %1" : "+v"(d0) : "v"(db));
            // a mix in the proportions of the four-lane tube wave: 14 packed, 10 DPP moves, 5 moves, 11 plain fp32, 3 int (per sample)
            if (KIND == 13) asm volatile("v_pk_fma_f32 %8, %8, %10, %10\n v_mov_b32_dpp %0, %1 row_ror:4 row_mask:0xf bank_mask:0x5\n v_fma_f32 %2, %2, %11, %12\n v_pk_fma_f32 %9, %9, %10, %10\n v_mov_b32 %3, %11\n v_mov_b32_dpp %4, %5 row_ror:4 row_mask:0xf bank_mask:0x5\n v_pk_mul_f32 %8, %8, %10\n v_mul_f32 %6, %6, %11" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+v"(p0), "+v"(p1) : "v"(pb), "v"(b), "v"(c));
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p1.x + p2.x + p3.x + p0.y + p1.y + p2.y + p3.y + (float)(d0 + d1 + d2 + d3) + (float)i0;
    sink[threadIdx.x] = s;
    if (lane == 0) { out[2 * w] = t0; out[2 * w + 1] = t1; }
}

static unsigned long long *dOut;
static float *dSink;
template <int KIND>
static void run(const char *name, int perGroup)
{
    printf("%-44s", name);
    for (int W = 1; W <= 4; W++) {
        hipLaunchKernelGGL(k<KIND>, dim3(1), dim3(256 * W), 0, 0, dOut, dSink);
        hipDeviceSynchronize();
        hipLaunchKernelGGL(k<KIND>, dim3(1), dim3(256 * W), 0, 0, dOut, dSink);
        hipDeviceSynchronize();
        unsigned long long h[32];
        hipMemcpy(h, dOut, sizeof h, hipMemcpyDeviceToHost);
        unsigned long long lo = ~0ull, hi = 0;
        for (int w = 0; w < 4 * W; w++) { if (h[2 * w] < lo) lo = h[2 * w]; if (h[2 * w + 1] > hi) hi = h[2 * w + 1]; }
        // the counter runs at 100 MHz on this part; the shader clock is read back from the ratio of a calibration loop
        printf("  W=%d %6.2f", W, (double)(hi - lo) / ((double)ITERS * REP * perGroup * W));
    }
    printf("\n");
}

int main()
{
    hipMalloc(&dOut, 64 * 8);
    hipMalloc(&dSink, 1024 * 4);
    printf("counter ticks per instruction per SIMD (W = waves per SIMD); multiply by (shader clock / counter clock)\n");
    run<4>("v_mov_b32 (8 independent)", 8);
    run<0>("v_fma_f32 (8 independent)", 8);
    run<8>("v_mul/add/sub_f32 (8 independent)", 8);
    run<1>("v_pk_fma_f32 (4 independent pairs)", 8);
    run<2>("v_fma_f64 (4 independent)", 8);
    run<3>("v_mov_b32_dpp row_ror (8-ring)", 8);
    run<5>("v_rcp_f32 (8 independent)", 8);
    run<6>("v_cndmask_b32 (8 independent)", 8);
    run<17>("v_cmp_gt_f32 + v_cndmask_b32 (4 pairs)", 8);
    run<15>("v_cndmask_b32_e64, mask in an SGPR pair", 8);
    run<14>("v_fma_f32 with an SGPR source", 8);
    run<16>("v_pk_fma_f32 with an SGPR-pair source", 8);
    run<7>("v_add_u32 (dependent)", 8);
    run<9>("v_fma_f32 dependent chain", 8);
    run<10>("v_pk_fma_f32 dependent chain", 8);
    run<12>("v_fma_f64 dependent chain", 8);
    run<11>("v_fma_f32 -> dpp mov -> v_fma_f32 chain", 8);
    run<13>("tube-wave mix (3 pk, 2 dpp, 1 mov, 2 plain)", 8);
    return 0;
}
