// trm_solo.hip -- one voice per lane, ONE WAVE per workgroup: every stage of the sample loop (TRMTubeModel.m:272-361)
// in the same wave, no barriers.
//
// trm_kernels.hip's trm_tube_kernel runs the stages as a pipeline of seven waves that meet at a barrier every two
// samples; at a saturating batch its SIMDs issue 72 % of what they could (profiles/valu_pmc_calibration_r04.txt), the rest
// is waves parked at the barrier behind whichever role got its issue slots last.  Here a wave owns its 64 voices outright:
//   per tube sample   tracks + oscillator -> FIR + mixing -> coefficients -> tube step, lane = voice, all state in VGPRs,
//                     nothing handed over through LDS but the tube-rate sample itself (a 64-sample ring per voice);
//   per 32 outputs    the converter, lane = OUTPUT TIME as in trm_tube_kernel: as soon as the ring holds the last tube sample
//                     a block of 32 outputs reads, the wave converts the block for its 64 voices (16 row pairs) and goes
//                     back to the sample loop.
// Two such waves share a SIMD (the kernel needs ~200 VGPRs), eight a CU; they are independent, so one's LDS and memory
// latencies are the other's issue slots.  LDS per wave: 17 KB of ring + 2.5 KB.
// One-shot and time-split launches (TubeArgs::seg_*); streaming stays with trm_tube_kernel.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "trm_devutil.h"
#include "trm_kernels.h"
#include "trm_lane.h"

namespace trm {

// the stages of a sample are kept apart in the instruction stream: interleaved (the scheduler's choice, for latency) their
// live ranges add up to more registers than two waves per SIMD have
#ifndef SOLO_FENCE
#define SOLO_FENCE __builtin_amdgcn_sched_barrier(0);
#endif

constexpr int kSoloRing = 64;                    // tube-rate samples kept per voice: a block's windows span at most 31 + 29 of them
constexpr int kSoloStride = kSoloRing + 4;       // multiple of 4 floats (16-byte aligned rows), odd multiple of 4 banks

// kDown: a down-sampling batch -- tube-rate samples go to HBM for trm_downsample_kernel, no converter here
template <bool kSeg, bool kDown>
__global__ __launch_bounds__(kWave) __attribute__((amdgpu_waves_per_eu(2, 2))) void trm_tube_kernel_s(const Const C, const TubeArgs A)
{
    if (A.gate && ((*A.gate != 0u) ? 1u : 0u) != A.gate_want) return;
    __shared__ __attribute__((aligned(16))) float sY[kWave * kSoloStride];   // tube-rate rings
    __shared__ uint2 sPtr[kWave];                                            // convert: PCM pointer per voice
    __shared__ uint32_t sNout[kWave];                                        // convert: outputs of this launch per voice
    __shared__ float sMx[kWave * 4];                                         // convert: running max |y| per (voice, column & 3)
    __shared__ float sNoise[kNoiseRing];
    __shared__ __attribute__((aligned(16))) float sFir[kFirTaps + 3];                   // the FIR's 49 taps, unfolded

    const int lane = threadIdx.x;
    const uint32_t wg = A.wg_base + blockIdx.x;
    const uint32_t seg = kSeg ? wg / A.seg_wg_per_seg : 0u;
    const uint32_t vblock = kSeg ? wg - seg * A.seg_wg_per_seg : wg;
    const uint32_t vRaw = vblock * kWave + lane;
    const bool laneValid = vRaw < A.nvoices;
    const uint32_t v = laneValid ? vRaw : A.nvoices - 1;
    const uint32_t CP = (uint32_t)C.controlPeriod;
    const uint32_t inc = C.timeRegisterIncrement;
    const uint32_t pad2 = 2u * (uint32_t)C.padSize;
    auto outputs_before = [&](uint64_t end) { return end == 0 ? 0u : (uint32_t)(((end << 16) - 1) / inc + 1); };
    auto seg_begin = [&](uint32_t sgm) { return sgm == 0 ? 0u : A.seg_first + (sgm - 1) * A.seg_periods; };

    // the frames this launch runs for this lane (trm_tube_kernel: same bookkeeping)
    const uint32_t nfrAll = min(A.nframes[v], A.max_nframes);
    uint32_t nfr = nfrAll, segFrame0 = 0, segOutEnd = 0;
    bool segLast = true;
    if (kSeg) {
        const uint32_t nper = nfrAll > 0 ? nfrAll - 1 : 0;
        const uint32_t pLo = seg_begin(seg), pEnd = seg_begin(seg + 1);
        segFrame0 = pLo > A.seg_warm ? pLo - A.seg_warm : 0u;
        if (seg > 0 && pLo >= nper) nfr = 0;
        else if (nfrAll > 0) {
            const uint32_t pHi = pEnd < nper ? pEnd : nper;
            nfr = pHi - segFrame0 + 1;
            segLast = pHi == nper;
            segOutEnd = outputs_before((uint64_t)pHi * CP);
        }
    }
    const uint32_t nfrMax = wave_max_u32(nfr);
    const uint32_t ntubeMax = nfrMax > 0 ? (nfrMax - 1) * CP : 0;
    const uint32_t nBase = kSeg ? segFrame0 * CP : 0u;
    const uint32_t kBase = kSeg ? outputs_before((uint64_t)seg_begin(seg) * CP) : 0u;
    const uint32_t nTotal = nfrMax > 0 ? ntubeMax + pad2 : 0;
    const float *frames = A.frames + (nfr > 0 ? (A.frame_offset[v] + segFrame0) * 16 : 0);
    const uint32_t ntubeLane = nfr > 0 ? (nfr - 1) * CP : 0;

    uint32_t noutLane = 0, noutAll = 0;
    if (nfr > 0) noutLane = (uint32_t)((((uint64_t)ntubeLane + pad2) * 65536ull + inc - 1) / inc);
    if (kSeg) {
        if (nfrAll > 0) noutAll = (uint32_t)((((uint64_t)(nfrAll - 1) * CP + pad2) * 65536ull + inc - 1) / inc);
        noutLane = nfr > 0 ? (segLast ? noutAll : segOutEnd) - kBase : 0u;
    }
    if (!laneValid) noutLane = 0;
    const uint32_t noutMax = wave_max_u32(noutLane);
    const uint32_t nBlocks = kDown ? 0u : (noutMax + kCvtCols - 1) / kCvtCols;

    for (int i = lane; i < kWave * kSoloStride; i += kWave) sY[i] = 0.0f;      // 25 zeros of pre-roll
    {
        const uintptr_t myOut = reinterpret_cast<uintptr_t>(A.out + A.out_offset[v] + (kSeg ? kBase : 0u));
        sPtr[lane] = make_uint2((uint32_t)myOut, (uint32_t)(myOut >> 32));
        sNout[lane] = noutLane;
        for (int i = 0; i < 4; i++) sMx[lane * 4 + i] = 0.0f;
        if (lane < kFirTaps + 3) sFir[lane] = lane < kFirTaps ? C.fir[lane < kFirUnique ? lane : (kFirTaps - 1) - lane] : 0.0f;
    }

    // ------------------------------------------------------------ the converter (lane = output time)
    const int col = lane & (kCvtCols - 1);
    const int ha = lane >= kCvtCols ? 1 : 0;          // which of a row's two voices
    uint32_t blk = 0;
    uint32_t needLast = 0;      // last tube sample (of this launch) the next block reads
    bool dmaPending = false;    // a noise request may still be in flight (uniform)
    auto block_need = [&](uint32_t b) {
        uint32_t need = src_position(kBase + b * kCvtCols + (kCvtCols - 1), inc) - nBase;
        return need < nTotal - 1 ? need : nTotal - 1;
    };
    if (nBlocks > 0) needLast = block_need(0);
    typedef __attribute__((address_space(1))) float *GlobalFloatPtr;
    typedef __attribute__((address_space(3))) float *LdsFloatPtr;
    auto convert_block = [&]() {
        const uint32_t kLane = blk * kCvtCols + col;
        const uint32_t e = src_position(kBase + kLane, inc);
        // the 16-byte aligned 32-sample window that holds the output's 26 (trm_tube_kernel), as 8 quads of a ring without a
        // mirror: quad q sits at (winBase + 4q) mod 64
        const uint32_t winBase = e & (kSoloRing - 1) & ~3u;
        const float *pc = A.src_rows + (size_t)src_phase(kBase + kLane, inc) * kSrcRowC - (e & 3u);
        v2f cc[15];
        for (int q = 0; q < 15; q++) cc[q] = v2f{pc[2 * q], pc[2 * q + 1]};
        uint32_t qoff[8];
        for (int q = 0; q < 8; q++) qoff[q] = ha * kSoloStride + ((winBase + 4u * q) & (kSoloRing - 1));
        const int mcol = col & 3;
#pragma unroll 2
        for (int r = 0; r < 32; r++) {
            const int va = 2 * r;                     // this row: voices va, va + 1
            const float *ra = &sY[va * kSoloStride];
            float4 qa[8];
            for (int q = 0; q < 8; q++) qa[q] = *reinterpret_cast<const float4 *>(ra + qoff[q]);
            const uint32_t na = sNout[va + ha];
            const uint2 pa = sPtr[va + ha];
            v2f a0 = v2f{qa[0].x, qa[0].y} * cc[0], a1 = v2f{qa[0].z, qa[0].w} * cc[1];
            for (int q = 1; q < 7; q++) {
                a0 = __builtin_elementwise_fma(v2f{qa[q].x, qa[q].y}, cc[2 * q], a0);
                a1 = __builtin_elementwise_fma(v2f{qa[q].z, qa[q].w}, cc[2 * q + 1], a1);
            }
            a0 = __builtin_elementwise_fma(v2f{qa[7].x, qa[7].y}, cc[14], a0);
            a0 += a1;
            const float ya = a0.x + a0.y;
            const bool okA = kLane < na;
            if (okA) reinterpret_cast<GlobalFloatPtr>(((uintptr_t)pa.y << 32) | pa.x)[kLane] = ya;
            __builtin_amdgcn_ds_fmaxf((LdsFloatPtr)&sMx[(va + ha) * 4 + mcol], okA ? fabsf(ya) : 0.0f, 0, 0, false);
        }
        dmaPending = false;     // the coefficient loads above came back: so has every noise request issued before them
        blk++;
        if (blk < nBlocks) needLast = block_need(blk);
    };

    // ------------------------------------------------------------ the sample loop (lane = voice)
    auto sine = [&](int i) { return sine_table(i); };
    OscState S;
    ExciteTrack T;
    FirState FS;
    CoefTrack CT;
    Waves wA, wB;
    TubeFilters F;
    S.oscPos = 0.0;
    if (kSeg) {
        const double *ph = A.seg_phase + vRaw;
        const size_t pitch = (size_t)A.seg_wg_per_seg * kWave;
        for (uint32_t q = 1; q <= seg; q++) {
            const double t = S.oscPos + ph[q * pitch];
            S.oscPos = t > 511.0 ? t - 512.0 : t;
        }
    }
    for (int i = 0; i < 24; i++) FS.fir[i] = 0.f;
    waves_reset(wA);
    waves_reset(wB);
    filters_reset(F);
    auto vcopy = [](float s) { float r; asm volatile("v_mov_b32 %0, %1" : "=v"(r) : "s"(s)); return r; };
    // The damping factor sits in a third of the tube step's instructions: a vector-register copy (an instruction with a scalar
    // operand issues at the slow rate on a shared SIMD).  The other constants are read a few times per sample and stay
    // scalar: as vector registers they would be sixteen more than two waves per SIMD leave room for.
    TubeConst TC;
    TC.damping = vcopy(C.damping); TC.mCoeff = C.mCoeff; TC.nCoeff = C.nCoeff;
    for (int i = 0; i < 4; i++) TC.nasalTd[i] = C.nasalTd[i];
    TC.nasalK6a = C.nasalK6a; TC.onePlusNK6 = C.onePlusNK6;
    TC.ta0 = C.ta0; TC.tb1 = C.tb1; TC.throatGain = C.throatGain;
    CoefConst CC;
    CC.damping = TC.damping; CC.apScaleSq = C.apScaleSq; CC.mA10 = C.mA10;
    CC.noseR1sq = C.noseR1sq; CC.invSampleRate = C.invSampleRate; CC.fricGain = 1.0f;

    const float *const lpNoise = A.lp_noise + nBase;
    auto fill_noise_half = [&](uint32_t nFirst, int half) { dma4(lpNoise + nFirst + lane, &sNoise[half * kNoiseHalf]); };
    if (nTotal > 0) {
        fill_noise_half(0, 0);
        fill_noise_half(kNoiseHalf, 1);
        dma_wait_all();
    }
    float *const ring = &sY[lane * kSoloStride];
    float *const tubeOut = kDown ? A.tube_out + A.tube_offset[v] : nullptr;
    const uint32_t tubeEnd = ntubeLane + (segLast ? pad2 : 0u);
    const uint32_t segStart = kSeg ? seg_begin(seg) * CP : 0u;
    auto frame_at = [&](uint32_t i) { return nfr > 0 ? (i < nfr ? i : nfr - 1) : 0u; };
    uint32_t j = CP, f = 0;
    auto sample = [&](const Waves &o, Waves &nw, uint32_t n) {
        if (j == CP) {          // -setControlRateParameters:previous: (TRMTubeModel.m:289)
            j = 0;
            f++;
            float prev[16], cur[16];
            load_frame(frames, frame_at(f - 1), prev, 4);
            load_frame(frames, frame_at(f), cur, 4);
            excite_track_setup(T, C, prev, cur);
            coef_track_setup(CT, C, prev, cur);
        }
        if ((n & (kNoiseHalf - 1)) == 0 && n > 0) {
            // entering a noise half: it was requested one half ago.  vmcnt counts the PCM stores too, in order: waiting here
            // would wait for the last block's stores -- but a converter block since the request has already waited for it
            if (dmaPending) dma_wait_all();
            fill_noise_half(n + kNoiseHalf, ((n / kNoiseHalf) + 1) & 1);
            dmaPending = true;
        }
        const OscOut O = osc_sample(S, T, C, (int)j, sine);
        SOLO_FENCE
        const Excitation E = mix_sample_unfolded(FS, C, sFir, O, sNoise[n & (kNoiseRing - 1)]);
        SOLO_FENCE
        const Coefs K = coef_sample<false>(CT, CC, (int)j);
        SOLO_FENCE
        j++;
        float y = tube_step(o, nw, F, TC, E, K);
        SOLO_FENCE
        y = n < ntubeLane ? y : 0.0f;      // zero flush / voices shorter than the group's longest
        ring[(nBase + n + (kSrcWindow - 1)) & (kSoloRing - 1)] = y;
        if (kDown) {
            const uint32_t gn = nBase + n;
            if (laneValid && gn >= segStart && n < tubeEnd) tubeOut[gn] = y;
        }
    };
    for (uint32_t n = 0; n < nTotal; n += 2) {
        // both samples are stepped even when the second lies past nTotal (its output is forced to 0 and never read)
        sample(wA, wB, n);
        sample(wB, wA, n + 1);
        while (blk < nBlocks && needLast < n + 2) convert_block();
    }
    while (blk < nBlocks) convert_block();
    dma_wait_all();     // nothing may still be writing LDS when the wave ends

    if (laneValid && !kDown) {
        const float4 m = *reinterpret_cast<const float4 *>(&sMx[lane * 4]);
        const float myMax = fmaxf(fmaxf(m.x, m.y), fmaxf(m.z, m.w));
        if (kSeg) {
            if (seg == 0) A.number_samples[vRaw] = noutAll;
            if (myMax > 0.0f) atomicMax(reinterpret_cast<unsigned int *>(&A.max_sample[vRaw]), __float_as_uint(myMax));
        } else {
            A.number_samples[vRaw] = noutLane;
            A.max_sample[vRaw] = myMax;
        }
    }
}

hipError_t launch_tube_solo(const Const &c, const TubeArgs &a, hipStream_t stream)
{
    if (a.nvoices == 0) return hipSuccess;
    if (a.stream_state) return hipErrorInvalidValue;
    const uint32_t grid = a.seg_periods ? a.seg_grid : (a.nvoices + kWave - 1) / kWave;
    TubeArgs s = a;
    s.wg_base = 0;
    const bool down = !c.upsample;
    if (down && !a.tube_out) return hipErrorInvalidValue;
    if (a.seg_periods) {
        if (down) hipLaunchKernelGGL((trm_tube_kernel_s<true, true>), dim3(grid), dim3(kWave), 0, stream, c, s);
        else hipLaunchKernelGGL((trm_tube_kernel_s<true, false>), dim3(grid), dim3(kWave), 0, stream, c, s);
    } else {
        if (down) hipLaunchKernelGGL((trm_tube_kernel_s<false, true>), dim3(grid), dim3(kWave), 0, stream, c, s);
        else hipLaunchKernelGGL((trm_tube_kernel_s<false, false>), dim3(grid), dim3(kWave), 0, stream, c, s);
    }
    return hipGetLastError();
}

int tube_solo_kernel_blocks_per_cu()
{
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, trm_tube_kernel_s<false, false>, kWave, 0) != hipSuccess) return -1;
    return n;
}

}  // namespace trm
