// trm_devutil.h -- device-side helpers shared by the tube kernels (trm_kernels.hip, trm_quad.hip).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <mutex>
#include <vector>

namespace trm {

// More than 64 KB of dynamic LDS has to be allowed once per kernel and device (hipFuncSetAttribute).  One of these per
// kernel instance (a function-local static of its launcher): handles of different devices launch from different
// threads, so the bookkeeping is locked, and it grows with the device index instead of capping it.
struct DynamicLdsAllowance {
    std::mutex m;
    std::vector<char> done;
    hipError_t ensure(const void *kernel, int bytes)
    {
        int dev = 0;
        hipError_t e = hipGetDevice(&dev);
        if (e != hipSuccess) return e;
        std::lock_guard<std::mutex> lock(m);
        if ((size_t)dev >= done.size()) done.resize((size_t)dev + 1, 0);
        if (!done[dev]) {
            e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
            if (e != hipSuccess) return e;
            done[dev] = 1;
        }
        return hipSuccess;
    }
};

constexpr int kWave = 64;
constexpr int kNoiseRing = 128;      // noise ring: one float per tube sample, refilled by halves of 64
constexpr int kNoiseHalf = 64;
// tube -> convert hand-off: per voice a ring of the last 128 tube-rate samples (+ mirror)
constexpr int kYRing = 128;
constexpr int kYMirror = 32;          // slots 0..31 repeated after the ring: a 32-sample aligned window never wraps
constexpr int kYStride = kYRing + kYMirror + 4;   // multiple of 4 floats: 16-byte aligned rows for ds_read_b128
constexpr int kCvtCols = 32;         // convert: outputs per block (lanes 0-31 / 32-63 = two voices)

// ---------------------------------------------------------------- LDS-DMA helpers
// global_load_lds_*: asynchronous global -> LDS copy, no VGPR destination.  The LDS address is a
// wave-uniform base (M0) + lane * size; the global source address is per lane.  Completion is
// tracked by vmcnt; the compiler does not know these writes, so readers wait explicitly.
typedef __attribute__((address_space(1))) const void *GlobalPtr;
typedef __attribute__((address_space(3))) void *LdsPtr;
typedef float v2f __attribute__((ext_vector_type(2)));   // two-wide fp32: v_pk_fma_f32 / v_pk_mul_f32

__device__ __forceinline__ void dma16(const float *src, float *ldsBaseUniform)
{
    __builtin_amdgcn_global_load_lds((GlobalPtr)src, (LdsPtr)ldsBaseUniform, 16, 0, 0);
}
__device__ __forceinline__ void dma4(const float *src, float *ldsBaseUniform)
{
    __builtin_amdgcn_global_load_lds((GlobalPtr)src, (LdsPtr)ldsBaseUniform, 4, 0, 0);
}
__device__ __forceinline__ void dma_wait_all() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// The pipeline's step barrier.  Only LDS traffic crosses it (every hand-off between the waves of a workgroup
// lives in LDS), so it waits for this wave's LDS operations and NOT for its global stores: __syncthreads()
// also drains vmcnt, which parks the converter waves (and with them the whole workgroup) behind the
// acknowledgement of PCM stores nobody in the workgroup reads.
__device__ __forceinline__ void step_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v)
{
    for (int off = 32; off > 0; off >>= 1) {
        uint32_t o = __shfl_xor(v, off, kWave);
        v = o > v ? o : v;
    }
    return __builtin_amdgcn_readfirstlane(v);
}

__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v)
{
    for (int off = 32; off > 0; off >>= 1) {
        uint32_t o = __shfl_xor(v, off, kWave);
        v = o < v ? o : v;
    }
    return __builtin_amdgcn_readfirstlane(v);
}

// Diagnostic build only (-DTRM_STAMP, tools/stage_profile.py): per-role cycles spent working vs
// waiting at the step barrier.  In the product build these macros expand to nothing.
#ifdef TRM_STAMP
#ifndef TRM_STAMP_LONG
#define TRM_STAMP_LONG 1600      /* cycles: a step's work above this counts as a long step */
#endif
/* per-step trace of workgroup 0 (tools/stage_profile.py --trace): work cycles of role r at step i in stamps[400000 + r * 4096 + i] */
#ifdef TRM_STAMP_TRACE
#define STAMP_TRACE if (blockIdx.x == 0 && lane == 0 && A.stamps && st_n < 4096) A.stamps[400000 + role * 4096 + st_n] = st_t1 - st_t0; st_n++;
#else
#define STAMP_TRACE
#endif
#define STAMP_DECL unsigned st_n = 0; (void)st_n; unsigned long long st_work = 0, st_wait = 0, st_t0 = 0, st_t1 = 0, st_long = 0, st_excess = 0, st_max = 0; const unsigned long long st_born = __builtin_amdgcn_s_memrealtime();
#define SUB_DECL unsigned long long sub_t = 0, sub_acc[6] = {0, 0, 0, 0, 0, 0};
#define SUB_START sub_t = __builtin_readcyclecounter();
#define SUB_LAP(i_) { unsigned long long n_ = __builtin_readcyclecounter(); sub_acc[i_] += n_ - sub_t; sub_t = n_; }
#define SUB_STORE(role_) if (lane == 0 && A.stamps) for (int i_ = 0; i_ < 6; i_++) A.stamps[(blockIdx.x * kStampRoles + (role_)) * 8 + 2 + i_] = sub_acc[i_];
#define STAMP_BEGIN st_t0 = __builtin_readcyclecounter();
#define STAMP_MID st_t1 = __builtin_readcyclecounter(); st_work += st_t1 - st_t0; STAMP_TRACE if (st_t1 - st_t0 > TRM_STAMP_LONG) { st_long++; st_excess += st_t1 - st_t0 - TRM_STAMP_LONG; } if (st_t1 - st_t0 > st_max) st_max = st_t1 - st_t0;
#define STAMP_END st_wait += __builtin_readcyclecounter() - st_t1;
#define STAMP_STORE(role_)                                                          \
    if (lane == 0 && A.stamps) {                                                    \
        A.stamps[(blockIdx.x * kStampRoles + (role_)) * 8] = st_work;                    \
        A.stamps[(blockIdx.x * kStampRoles + (role_)) * 8 + 1] = st_wait;                \
        A.stamps[(blockIdx.x * kStampRoles + (role_)) * 8 + 2] = __builtin_amdgcn_s_getreg((31 << 11) | 4); /* HW_ID */ \
        A.stamps[(blockIdx.x * kStampRoles + (role_)) * 8 + 3] = st_born;   /* 100 MHz wall clock: when the wave entered its loop ... */ \
        A.stamps[(blockIdx.x * kStampRoles + (role_)) * 8 + 4] = __builtin_amdgcn_s_memrealtime();   /* ... and left it */ \
        A.stamps[(blockIdx.x * kStampRoles + (role_)) * 8 + 5] = st_long;                \
        A.stamps[(blockIdx.x * kStampRoles + (role_)) * 8 + 6] = st_excess;              \
        A.stamps[(blockIdx.x * kStampRoles + (role_)) * 8 + 7] = st_max;                 \
    }
#else
#define STAMP_DECL
#define SUB_DECL
#define SUB_START
#define SUB_LAP(i_)
#define SUB_STORE(role_)
#define STAMP_BEGIN
#define STAMP_MID
#define STAMP_END
#define STAMP_STORE(role_)
#endif

__device__ __forceinline__ void load_frame(const float *frames, uint32_t fi, float *dst, int quads)
{
    const float4 *p = reinterpret_cast<const float4 *>(frames + (size_t)fi * 16);
    for (int q = 0; q < quads; q++) {
        float4 x = p[q];
        dst[4 * q] = x.x; dst[4 * q + 1] = x.y; dst[4 * q + 2] = x.z; dst[4 * q + 3] = x.w;
    }
}

}  // namespace trm
