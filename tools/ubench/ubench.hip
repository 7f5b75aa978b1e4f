// Instruction-cost microbenchmark for gfx950: cycles per wave-instruction for ONE wave on a SIMD
// (the regime the TRM pipeline runs in) and for two waves sharing a SIMD.  Diagnostic tool only.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

#define REP 64
#define ITERS 200

template <int KIND>
__global__ void k(unsigned long long *out, float *sink, int nwaves)
{
    __shared__ float lds[4096];
    const int lane = threadIdx.x & 63;
    const int w = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = (float)i;
    __syncthreads();
    float a[8], b = 1.0001f + lane * 1e-6f, c = 0.5f;
    double da[4], db = 1.0000001 + lane * 1e-9;
    typedef float v2f __attribute__((ext_vector_type(2)));
    v2f pa[8], pb = {b, b + 1e-3f};
    for (int i = 0; i < 8; i++) { a[i] = (float)i + lane; pa[i] = v2f{a[i], a[i] + 1.f}; }
    for (int i = 0; i < 4; i++) da[i] = (double)i + lane;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int r = 0; r < REP / 8; r++) {
#pragma unroll
            for (int i = 0; i < 8; i++) {
                if (KIND == 0) a[i] = __builtin_fmaf(a[i], b, c);                       // independent chains of 8
                if (KIND == 1) pa[i] = __builtin_elementwise_fma(pa[i], pb, pb);        // v_pk_fma_f32
                if (KIND == 2) da[i & 3] = __builtin_fma(da[i & 3], db, db);            // v_fma_f64
                if (KIND == 3) da[i & 3] = da[i & 3] + db;                              // v_add_f64
                if (KIND == 4) a[i] = __builtin_amdgcn_rcpf(a[i]);                      // v_rcp_f32
                if (KIND == 5) a[i] = a[i] > c ? a[i] - b : a[i];                       // cmp + sub + cndmask
                if (KIND == 6) a[0] = __builtin_fmaf(a[0], b, c);                       // ONE dependent chain
                if (KIND == 7) a[i] += lds[(lane * 4 + i * 256 + r * 16) & 4095];        // ds_read_b32 + add
                if (KIND == 8) {                                                        // ds_read_b128 + 4 adds
                    float4 q = *reinterpret_cast<float4 *>(&lds[(lane * 4 + i * 256 + r * 16) & 4092]);
                    a[i] += q.x + q.y + q.z + q.w;
                }
                if (KIND == 9) asm volatile("v_mov_b32 %0, %0" : "+v"(a[i]));            // v_mov
                if (KIND == 10) asm volatile("s_nop 0");
                if (KIND == 11) a[i] = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, a[i]), __builtin_bit_cast(int, a[(i + 1) & 7]), 0x124, 0xF, 0x5, false));   // v_mov_b32_dpp row_ror:4, bank-masked merge
                if (KIND == 12) {   // fma then a DPP merge of its result (the hazard the quad tube step lives on)
                    a[i] = __builtin_fmaf(a[i], b, c);
                    a[(i + 4) & 7] = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, a[(i + 4) & 7]), __builtin_bit_cast(int, a[i]), 0x128, 0xF, 0x2, false));
                }
                if (KIND == 13) {   // scalar fma, explicitly not packed
                    asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
                }
                if (KIND == 14) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b));
                if (KIND == 15) asm volatile("s_add_u32 s20, s20, 1" ::: "s20");
            }
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int i = 0; i < 8; i++) s += a[i] + pa[i].x + pa[i].y;
    for (int i = 0; i < 4; i++) s += (float)da[i];
    sink[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (lane == 0) out[blockIdx.x * 8 + w] = t1 - t0;
}

template <int KIND>
void run(const char *name, int threads, unsigned long long *dOut, float *dSink)
{
    hipLaunchKernelGGL(k<KIND>, dim3(1), dim3(threads), 0, 0, dOut, dSink, threads / 64);
    hipDeviceSynchronize();
    unsigned long long h[8];
    hipMemcpy(h, dOut, sizeof h, hipMemcpyDeviceToHost);
    printf("%-28s threads %3d: %.2f cycles per wave-instruction-slot (wave0), wave4 %.2f\n", name, threads,
           (double)h[0] / (ITERS * REP), threads > 256 ? (double)h[4] / (ITERS * REP) : 0.0);
}

int main()
{
    unsigned long long *dOut;
    float *dSink;
    hipMalloc(&dOut, 64 * 8);
    hipMalloc(&dSink, 1024 * 4);
    for (int threads : {64, 256, 512}) {       // 1 wave; 1 per SIMD; 2 per SIMD
        run<0>("v_fma_f32 (8 chains)", threads, dOut, dSink);
        run<1>("v_pk_fma_f32 (8 chains)", threads, dOut, dSink);
        run<2>("v_fma_f64 (4 chains)", threads, dOut, dSink);
        run<3>("v_add_f64 (4 chains)", threads, dOut, dSink);
        run<4>("v_rcp_f32", threads, dOut, dSink);
        run<5>("cmp+sub+cndmask (3 instr)", threads, dOut, dSink);
        run<6>("v_fma_f32 (1 dependent chain)", threads, dOut, dSink);
        run<7>("ds_read_b32 + v_add", threads, dOut, dSink);
        run<8>("ds_read_b128 + 4 v_add", threads, dOut, dSink);
        run<9>("v_mov_b32", threads, dOut, dSink);
        run<10>("s_nop 0", threads, dOut, dSink);
        run<11>("v_mov_b32_dpp merge", threads, dOut, dSink);
        run<12>("v_fma + dpp merge (2 instr)", threads, dOut, dSink);
        run<13>("v_fma_f32 (asm, unpacked)", threads, dOut, dSink);
        run<14>("v_cndmask_b32", threads, dOut, dSink);
        run<15>("s_add_u32", threads, dOut, dSink);
    }
    return 0;
}
