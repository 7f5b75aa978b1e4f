// trm_quad_dev.h -- device-side pieces shared by the workgroup-pipeline kernels with several lanes per voice
// (trm_quad.hip: four lanes per voice, 16 voices per workgroup; trm_oct.hip: eight lanes, 8 voices): the LDS rings'
// geometry, the hand-off depths and the two-word hand-shake between the mix and convert waves.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "trm_devutil.h"
#include "trm_lane.h"
#include "trm_quad.h"

namespace trm {

// The two LDS words through which the mix and convert waves hand coefficient rows to each other outside the step
// barrier.  Publishing = LDS-only release fence (every lane: the rows were stored by all of them), then the flag;
// consuming = the flag, then an LDS-only acquire fence.  "local": the fences order LDS traffic only -- a full
// workgroup-scope release would also drain the convert wave's PCM stores (vmcnt), which nobody here reads.
__device__ __forceinline__ void lds_flag_publish(uint32_t *flag, uint32_t value, bool writer)
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    if (writer) __hip_atomic_store(flag, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ uint32_t lds_flag_consume(uint32_t *flag)
{
    const uint32_t v = __builtin_amdgcn_readfirstlane(__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
    return v;
}

constexpr int kORing = 64;           // osc -> mix ring: (a, b) per tube sample
constexpr int kOMirror = 32;         // slots 0..31 repeated after the ring: a 26-sample window never wraps
constexpr int kOStride = kORing + kOMirror + 4;   // + 32 bytes: a row of lanes (4 voices x 2 distinct window starts) reads 8 different 16-byte bank groups
// Hand-off depths in steps (x kSub blocks).  Block b of the mix wave's records lives in buffer b % (4 kSub): written
// during step j+1, scanned during j+2, read by the tube during j+4 (its head during j+3).  The coefficient waves'
// records live in buffer b % (3 kSub): written during step j+2, read during j+4 (head: j+3).
constexpr int kXDepth = 4, kKDepth = 3;
constexpr int kRowBufs = 3;          // converter coefficient rows staged in LDS: block B in buffer B % 3
constexpr int kRowPitch = kSrcRowC + 4;  // staged coefficient rows: 144 bytes apart, so that 16 lanes reading 16 rows hit 16 bank groups
constexpr int kQLead = 28;           // tube sample n sits at converter-ring slot (n + 28) & 127: the converter's 25 zeros of
                                     // pre-roll (TRMSampleRateConverter.m:138-150) + 3, so that a block of 4 is 16-byte aligned

// The oscillator FIR of one tube sample in direct form (fir_direct, trm_quad.h) over the LDS ring of oscillator reads: the
// 26-sample window as 13 aligned 16-byte reads of (a, b) pairs, fir_direct's four partial sums as two packed ones
// (even / odd window slots x (a, b)); `cab` = the window taps of the sample's parity as (a, b) pairs.
__device__ __forceinline__ float fir_window_dot(const float4 *wp, const v2f *cab)
{
    v2f acc0, acc1;
    {
        const float4 x = wp[0];
        acc0 = v2f{x.x, x.y} * cab[0];
        acc1 = v2f{x.z, x.w} * cab[1];
    }
#pragma unroll
    for (int q = 1; q < kFirWin / 2; q++) {
        const float4 x = wp[q];
        acc0 = __builtin_elementwise_fma(v2f{x.x, x.y}, cab[2 * q], acc0);
        acc1 = __builtin_elementwise_fma(v2f{x.z, x.w}, cab[2 * q + 1], acc1);
    }
    acc0 += acc1;
    return acc0.x + acc0.y;
}

// One converter output (src_dot32, trm_lane.h): the 32-term dot product of a 16-byte aligned window of the tube-rate ring
// with the phase's shifted coefficient row, as packed FMAs: (even, odd) partial sums, two chains.
__device__ __forceinline__ float cvt_window_dot(const float4 *w, const v2f *cc)
{
    float4 q[8];
    for (int i = 0; i < 8; i++) q[i] = w[i];
    v2f a0 = v2f{q[0].x, q[0].y} * cc[0];
    v2f a1 = v2f{q[0].z, q[0].w} * cc[1];
    for (int i = 1; i < 7; i++) {
        a0 = __builtin_elementwise_fma(v2f{q[i].x, q[i].y}, cc[2 * i], a0);
        a1 = __builtin_elementwise_fma(v2f{q[i].z, q[i].w}, cc[2 * i + 1], a1);
    }
    a0 = __builtin_elementwise_fma(v2f{q[7].x, q[7].y}, cc[14], a0);      // (terms 30, 31 are always zeros: 26 coefficients shifted by at most 3)
    a0 += a1;
    return a0.x + a0.y;
}

}  // namespace trm
