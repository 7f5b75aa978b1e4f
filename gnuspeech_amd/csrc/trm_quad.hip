// trm_quad.hip -- trm_tube_kernel_q: -[TRMTubeModel synthesize] (TRMTubeModel.m:272-361) for batches too
// small to fill the chip with one voice per lane.
//
// A batch of V voices has V serial recurrences of ~20 k tube samples per second of speech; with one voice
// per lane (trm_kernels.hip) 4096 voices are 64 workgroups on 64 of 256 CUs and every sample costs a full
// pass of five instruction streams.  Here a workgroup carries 16 voices and every voice owns FOUR lanes:
//   osc, mix, coef   lanes = 4 consecutive tube samples of a voice (the stages are feed-forward in time; the
//                    oscillator phase is a 4-wide prefix sum, the FIR runs in direct form over an LDS ring)
//   tube             lanes = 4 parts of the tube, junction values crossing a part boundary move by DPP
//   convert          lane = output time, as in the wide kernel (rows of 32 outputs x 2 voices)
// so one pass of the instruction streams advances 4 tube samples (a "block") and 256 workgroups cover 4096
// voices.  One barrier per STEP of kSub blocks.  kSub = 2 (batches of at most one workgroup per CU): the feed-forward
// waves run the step's two blocks as two independent instruction streams (each hides the other's latencies), the
// tube wave 8 samples in a row; 110 KB of LDS.  kSub = 1 (larger batches): half the hand-off depth, 76 KB of LDS, so
// that TWO workgroups share a CU and fill each other's issue slots.  At step i osc works on the blocks of step i, mix
// on those of step i-1, the two coefficient waves and the band-pass and throat scans (recurrences that only FEED
// the tube; they live in feed-forward waves) on those of step i-2, tube on those of step i-4, convert on whatever
// is complete, metered.  trm_tube_kernel_q<true, .> is the streaming instance (state in / out).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "trm_devutil.h"
#include "trm_kernels.h"
#include "trm_lane.h"
#include "trm_quad.h"
#include "trm_quad_dev.h"

// Timing experiments (tools/bench_variants.sh) live behind ONE switch; the product build defines none of them.
#ifndef TRM_EXPERIMENTS
#undef TRM_ABL_CVT
#undef TRM_ABL_SKIP
#undef TRM_QROLE_PERM
#undef TRM_QPRIO_TUBE
#undef TRM_QPRIO_OSC
#undef TRM_QPRIO_MIX
#undef TRM_QPRIO_COEF
#undef TRM_QPRIO_COEF2
#undef TRM_QPRIO_CVT
#undef TRM_THROAT_IN_OSC
#endif
#ifndef TRM_ABL_CVT
#define TRM_ABL_CVT 0
#endif
/* where the throat low-pass scan runs: the oscillator wave (1) or the mix wave (0) */
#ifndef TRM_THROAT_IN_OSC
#define TRM_THROAT_IN_OSC 1
#endif

namespace trm {

constexpr int kQV = 16;              // voices per workgroup
constexpr int kQB = kSlots;          // tube samples per block = time slots per voice
constexpr int kQRoles = 6;           // osc, mix, coef x2 (area | frication), tube, convert
constexpr int kKPitch = 2 * kWave + 4;   // coef -> tube: float4s per (buffer, sample): {kk | tp} x the tube wave's 64 lanes,
                                         // + 64 bytes so that the writers' four time slots fall in different LDS banks
constexpr int kXPitch = kQV + 4;         // mix / coef -> tube: float4s per (buffer, sample) of the per-voice records, same idea
// One-shot instances: the control frames are staged in LDS by LDS-DMA (the area wave, which has no other vector-memory
// traffic: one 1 KB transfer per control period = 16 voices x 64 bytes) and read by the three waves that turn them into
// tracks when a period starts -- see trm_oct.hip.  Frame f lives in slot f % 4: when the oscillator wave enters the
// period between frames p and p+1, the stager retires the transfer of frame p+2 (sent for a period earlier) and, one
// step later -- when the lagging coefficient waves are through with frame p-1 --, sends for frame p+3 in its place.  A
// control period must hold three steps (launch_tube_quad).  Carrying the prefetched frames in registers costs every
// coefficient wave ~30 register-to-register copies per step (the streaming instance, which must also run control periods
// of a few samples, still does).
constexpr int kQFrameRing = 4;
// kStream: the launch is a chunk of a streamed utterance (state restored / saved); a compile-time switch so that
// the one-shot instance carries none of it.
// LDS layout of one workgroup, carved out of ONE dynamically sized array: with static __shared__ arrays the compiler
// knows the size, concludes that two workgroups of six waves are "three waves per SIMD" and pads the kernel's VGPR
// allocation to 129 registers on purpose (the smallest count that keeps occupancy at three) -- but six waves on four
// SIMDs put FOUR waves of two co-resident workgroups on one SIMD whenever both start on the same one, 4 x 136
// registers do not fit 512, and the second workgroup waits for the first to end (measured: 256 workgroups at 0 us, 256
// at 1030 us).  With the size hidden and four waves per SIMD asked for, the allocation is the 122 registers the code uses.
template <int kSub>
struct QuadLds {
    static constexpr int kXBufs = kXDepth * kSub, kKBufs = kKDepth * kSub, kABufs = 2 * kSub;
    static constexpr size_t oO = 0;                                                       // float2 [kQV * kOStride]
    static constexpr size_t oA = oO + sizeof(float2) * kQV * kOStride;                    // float2 [kABufs * kWave]
    static constexpr size_t oX = oA + sizeof(float2) * kABufs * kWave;                    // float4 [kXBufs * kQB * kXPitch]
    static constexpr size_t oBP = oX + sizeof(float4) * kXBufs * kQB * kXPitch;           // float4 [kKBufs * kQB * kXPitch]
    static constexpr size_t oK = oBP + sizeof(float4) * kKBufs * kQB * kXPitch;           // float4 [kKBufs * kQB * kKPitch]
    static constexpr size_t oY = oK + sizeof(float4) * kKBufs * kQB * kKPitch;            // float  [kQV * kYStride]
    static constexpr size_t oRows = oY + sizeof(float) * kQV * kYStride;                  // float  [kRowBufs * kCvtCols * kRowPitch]
    static constexpr size_t oInfo = oRows + sizeof(float) * kRowBufs * kCvtCols * kRowPitch;   // uint4 [kQV]
    static constexpr size_t oMx = oInfo + sizeof(uint4) * kQV;                            // float  [8 * kWave]
    static constexpr size_t oNoise = oMx + sizeof(float) * 8 * kWave;                     // float  [kNoiseRing]
    static constexpr size_t oSync = oNoise + sizeof(float) * kNoiseRing;                  // uint32 [2]
    static constexpr size_t oFrames = oSync + 16;                                         // float4 [kQFrameRing * kQV * 4] (one-shot instances)
    static constexpr size_t kBytes = oFrames + sizeof(float4) * kQFrameRing * kQV * 4;
    static_assert(oA % 16 == 0 && oX % 16 == 0 && oBP % 16 == 0 && oK % 16 == 0 && oY % 16 == 0 && oRows % 16 == 0 && oInfo % 16 == 0, "16-byte aligned pieces");
};

// kSub: blocks per pipeline step (see the top of the file).
// kSeg (with kStream): a TIME-SPLIT launch (trm_kernels.h, TubeArgs::seg_*; round 4): the streaming instance's time bases,
// per workgroup -- workgroup w runs segment w / seg_wg_per_seg of 16 voices from rest, a warm-up ahead --, no state in or
// out.  What the one-voice-per-lane kernel's segment instance is for batches that fill the chip, this one is for a handful of
// voices: a single utterance becomes sixteen lanes' worth of segments on one CU.
template <bool kStream, int kSub, bool kSeg = false>
__global__ __launch_bounds__(kWave *kQRoles, 4) void trm_tube_kernel_q(const Const C, const TubeArgs A)
{
    static_assert(!kSeg || kStream, "the segment instance is built on the streaming instance");
    if (kSeg && A.gate && ((*A.gate != 0u) ? 1u : 0u) != A.gate_want) return;      // (two launches, the device runs one: TubeArgs::gate)
    constexpr int kStepN = kQB * kSub;       // tube samples per step
    typedef QuadLds<kSub> L;
    constexpr int kXBufs = L::kXBufs, kKBufs = L::kKBufs, kABufs = L::kABufs;
    extern __shared__ __attribute__((aligned(16))) unsigned char sLds[];
    float2 *const sO = reinterpret_cast<float2 *>(sLds + L::oO);           // osc -> mix: oscillator reads
    float2 *const sA = reinterpret_cast<float2 *>(sLds + L::oA);           // osc -> mix: {ax, ah1} per (block % kABufs, lane)
    float4 *const sX = reinterpret_cast<float4 *>(sLds + L::oX);           // mix -> tube: {gin, sig, thr} [buf][slot][voice]
    float4 *const sBP = reinterpret_cast<float4 *>(sLds + L::oBP);         // area -> tube: {C8, NC6, 1 + C8, 1 + NC6} [buf][slot][voice]
    float4 *const sK = reinterpret_cast<float4 *>(sLds + L::oK);           // coef -> tube: part records {transmission | injection}
    float *const sY = reinterpret_cast<float *>(sLds + L::oY);             // tube-rate rings
    float *const sRows = reinterpret_cast<float *>(sLds + L::oRows);       // mix -> convert: coefficient rows of 3 blocks
    uint4 *const sInfo = reinterpret_cast<uint4 *>(sLds + L::oInfo);
    float *const sMx = reinterpret_cast<float *>(sLds + L::oMx);
    float *const sNoise = reinterpret_cast<float *>(sLds + L::oNoise);
    uint32_t *const sRowSync = reinterpret_cast<uint32_t *>(sLds + L::oSync);   // [0] convert -> mix: first block whose staged rows are still needed; [1] mix -> convert: blocks staged
    float4 *const sF = reinterpret_cast<float4 *>(sLds + L::oFrames);          // control frames [f % kQFrameRing][voice][quarter]
    constexpr bool kLdsFrames = !kStream;

    constexpr int kStampRoles = kQRoles;
    (void)kStampRoles;
    const int lane = threadIdx.x & (kWave - 1);
    // wave -> role: waves w and w+4 share a SIMD; the tube wave (the only serial one) gets a SIMD to itself, the
    // others are paired so that the SIMDs carry about the same work (tools/stage_profile.py)
    const int waveIdx = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
#ifndef TRM_QROLE_PERM
#define TRM_QROLE_PERM 1, 5, 0, 4, 2, 3      /* mix convert osc tube | coef-area coef-fric: the frication wave (the longest
                                                feed-forward role) shares its SIMD with the converter (the shortest), the
                                                oscillator has one to itself like the tube (profiles/role_perm_r02.txt) */
#endif
    const int rolePerm[kQRoles] = {TRM_QROLE_PERM};
    int role = 0;
    for (int i = 0; i < kQRoles; i++) role = waveIdx == i ? rolePerm[i] : role;

    // lane -> (voice, slot/part): a row of 16 lanes = 4 banks (slot/part) x 4 voices
    const int part = (lane >> 2) & 3;
    const int vq = (lane >> 4) * 4 + (lane & 3);        // voice within the workgroup
    // time-split: workgroup -> (segment, block of 16 voices)
    uint32_t seg = 0, vblock = blockIdx.x;
    if (kSeg) {
        if (A.seg_map) {                     // (the pairs with work first: trm_kernels.hip, trm_seg_map_kernel)
            const uint2 m = A.seg_map[blockIdx.x];
            seg = m.x;
            vblock = m.y;
        } else {
            seg = blockIdx.x / A.seg_wg_per_seg;
            vblock = blockIdx.x - seg * A.seg_wg_per_seg;
        }
    }
    const uint32_t vRaw = vblock * kQV + vq;
    const bool laneValid = vRaw < A.nvoices;
    const uint32_t v = laneValid ? vRaw : A.nvoices - 1;
    const uint32_t CP = (uint32_t)C.controlPeriod;
    const uint32_t inc = C.timeRegisterIncrement;
    auto outputs_before = [&](uint64_t end) { return end == 0 ? 0u : (uint32_t)(((end << 16) - 1) / inc + 1); };
    auto seg_begin = [&](uint32_t sgm) { return sgm == 0 ? 0u : A.seg_first + (sgm - 1) * A.seg_periods; };

    const uint32_t nfrAll = min(A.nframes[v], A.max_nframes);
    // the frames this launch runs for this lane: the utterance's (chunk's), or those of the workgroup's segment with its warm-up
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
    // streaming: this launch is one chunk of a longer utterance (trm_kernels.h); one-shot = first and last at once
    constexpr bool streaming = kStream;
    constexpr bool saving = kStream && !kSeg;        // state out at the chunk's last sample (a segment starts from rest and leaves nothing)
    const bool sFirst = !streaming || kSeg || (A.stream_flags & 1u), sLast = !streaming || kSeg || (A.stream_flags & 2u);
    // TRAcT's loop (Applications/TRAcT/tube.c:1121-1136) reads the parameter set every sample and never interpolates: a
    // control period then runs on the frame that ENDS it, held (trm_stream_set_mode)
    const bool sHold = streaming && !kSeg && (A.stream_flags & 4u);
    const uint32_t nBase = kSeg ? segFrame0 * CP : streaming ? A.stream_n_base : 0u;
    const uint32_t kBase = kSeg ? outputs_before((uint64_t)seg_begin(seg) * CP) : streaming ? A.stream_k_base : 0u;
    float *const st = saving ? A.stream_state + (size_t)v * kStreamFloats : nullptr;
    // outputs of this launch for this lane's voice (segments: its own stretch; the voice's last segment runs to the utterance's end)
    uint32_t noutSeg = 0, noutAll = 0;
    if (kSeg) {
        if (nfrAll > 0) noutAll = (uint32_t)((((uint64_t)(nfrAll - 1) * CP + 2ull * (uint32_t)C.padSize) * 65536ull + inc - 1) / inc);
        noutSeg = (nfr > 0 && laneValid) ? (segLast ? noutAll : segOutEnd) - kBase : 0u;
    }
    // tube samples the tube stage produces: the utterance (chunk), then the converter's 2*pad zero flush (TRMRingBuffer.m:85-93)
    const uint32_t nTotal = nfrMax > 0 ? ntubeMax + (sLast ? 2u * (uint32_t)C.padSize : 0u) : 0;
    // the tube stage steps the blocks of step i-4 at step i; the convert wave finishes what is queued after the last barrier
    const uint32_t nSteps = nTotal > 0 ? (nTotal + kStepN - 1) / kStepN + 5 : 0;
    const float *frames = A.frames + (nfr > 0 ? (A.frame_offset[v] + segFrame0) * 16 : 0);
    const uint32_t ntubeLane = nfr > 0 ? (nfr - 1) * CP : 0;
    const uint32_t ntubeMin = wave_min_u32(ntubeLane);      // every voice of the group is still sounding below this
    auto frame_index = [&](uint32_t i) { return nfr > 0 ? (i < nfr ? i : nfr - 1) : 0u; };
    // staged frames (kLdsFrames): this lane's voice's frame f (nominal index: past the voice's last frame the stager
    // repeats it), `quads` 16-byte quarters of it
    auto frame_to = [&](uint32_t f, float *dst, int quads) {
        for (int q = 0; q < quads; q++) {
            const float4 x = sF[((f % kQFrameRing) * kQV + vq) * 4 + q];
            dst[4 * q] = x.x; dst[4 * q + 1] = x.y; dst[4 * q + 2] = x.z; dst[4 * q + 3] = x.w;
        }
    };
    // the stager (area wave): lane -> (voice lane / 4, quarter lane % 4) of the workgroup's 16 frames
    const float *stageSrc = nullptr;
    uint32_t stageNfr = 0;
    if (kLdsFrames && role == 2) {
        const uint32_t sv = min(vblock * kQV + ((uint32_t)lane >> 2), A.nvoices - 1);
        stageNfr = min(A.nframes[sv], A.max_nframes);
        stageSrc = A.frames + (stageNfr > 0 ? A.frame_offset[sv] * 16 : 0) + (lane & 3) * 4;
    }
    auto stage_frame = [&](uint32_t f) {
        const uint32_t fi = stageNfr > 0 ? (f < stageNfr ? f : stageNfr - 1) : 0u;
        dma16(stageSrc + (size_t)fi * 16, reinterpret_cast<float *>(&sF[(f % kQFrameRing) * kQV * 4]));
    };

    for (int i = threadIdx.x; i < kQV * kYStride; i += kWave * kQRoles) sY[i] = 0.0f;
    for (int i = threadIdx.x; i < kQV * kOStride; i += kWave * kQRoles) sO[i] = make_float2(0.0f, 0.0f);
    if (threadIdx.x < 2) sRowSync[threadIdx.x] = 0u;
    if (kLdsFrames && role == 2 && nSteps > 0) {
        stage_frame(0); stage_frame(1); stage_frame(2); stage_frame(3);
        dma_wait_all();
    }
    if (streaming && !sFirst) {
        // the last 32 tube samples of the previous chunk: positions -32 .. -1 of this chunk's rings
        __syncthreads();
        for (int i = threadIdx.x; i < kQV * 32; i += kWave * kQRoles) {
            const int q = i >> 5, t = i & 31;
            const uint32_t vv = vblock * kQV + q < A.nvoices ? vblock * kQV + q : A.nvoices - 1;
            const float *h = A.stream_state + (size_t)vv * kStreamFloats;
            sO[q * kOStride + 32 + t] = make_float2(h[8 + 2 * t], h[9 + 2 * t]);            // slot (-32 + t) & 63
            const uint32_t slot = (uint32_t)(t - 32 + kQLead) & (kYRing - 1);
            sY[q * kYStride + slot] = h[72 + t];
            if (slot < (uint32_t)kYMirror) sY[q * kYStride + slot + kYRing] = h[72 + t];
        }
    }
    __syncthreads();

    // The two recurrences that only FEED the tube (frication band-pass, throat low-pass: nothing of the tube's state
    // enters them) run in a feed-forward wave, serially over a voice's four slots: slot s's output is slot s+1's y1
    // and slot s+2's y2; what wraps around belongs to the next block.  A lane's evaluation is final from its own turn
    // on (its inputs no longer change), so the value of the last turn is everyone's.
    struct ScanState { float by1, by2, ny1, ny2, prevSig, thY, thNext; };
    auto scans_restore = [&](ScanState &Z, bool bandpass, bool throat) {
        Z.by1 = Z.by2 = Z.ny1 = Z.ny2 = Z.prevSig = Z.thY = Z.thNext = 0.0f;
        if (streaming && !sFirst) {
            if (bandpass) {     // st[4], st[5] = the band-pass outputs of positions -2, -1; st[6], st[7] = its inputs there
                Z.by1 = st[5];                                  // slot 0's y1
                Z.by2 = part == 0 ? st[4] : st[5];              // slot 0's y2, slot 1's y2
                Z.prevSig = part == 2 ? st[6] : st[7];          // slots 2, 3 of the "previous block"
            }
            if (throat) Z.thY = st[2];
        }
    };
    // band-pass (TRMFilters.m:19-29) of block `blk`: input sX.y (mix wave, in LDS), coefficients bp = this lane's
    // sample's {2 alpha, 2 beta, 2 gamma}; returns the lane's band-pass output
    auto bandpass_scan = [&](ScanState &Z, uint32_t blk, const float4 bp) {
        const float sig = reinterpret_cast<const float *>(&sX[((blk % kXBufs) * kQB + part) * kXPitch + vq])[1];
        const float X = q_take<0, kPart2 | kPart3>(sig, Z.prevSig);         // slots 0, 1 look into the previous block
        const float x2 = q_take<2, kPartAll>(X, X);
        float f = 0.0f;
#pragma unroll
        for (int t = 0; t < kQB; t++) {
            f = bandpass_eval<float>(bp.x, bp.y, bp.z, sig, x2, Z.by1, Z.by2);
            if (t == 0) { Z.by1 = q_take<1, kPart1>(Z.by1, f); Z.by2 = q_take<2, kPart2>(Z.by2, f); }
            if (t == 1) { Z.by1 = q_take<1, kPart2>(Z.by1, f); Z.by2 = q_take<2, kPart3>(Z.by2, f); }
            if (t == 2) { Z.by1 = q_take<1, kPart3>(Z.by1, f); Z.ny2 = q_take<2, kPart0>(Z.ny2, f); }
            if (t == 3) { Z.ny1 = q_take<1, kPart0>(Z.ny1, f); Z.ny2 = q_take<2, kPart1>(Z.ny2, f); }
        }
        Z.by1 = q_take<0, kPart0>(Z.by1, Z.ny1);
        Z.by2 = q_take<0, kPart0 | kPart1>(Z.by2, Z.ny2);
        Z.prevSig = sig;
        if (saving) {
            const uint32_t n = blk * kQB + (uint32_t)part;
            if (n + 2u == ntubeLane) { st[4] = f; st[6] = sig; }
            if (n + 1u == ntubeLane) { st[5] = f; st[7] = sig; }
        }
        return f;
    };
    // throat low-pass (:341, TRMFilters.m:72-77) over this lane's input `thr` of sample m
    auto throat_scan = [&](ScanState &Z, float thr, uint32_t m) {
        float ty = 0.0f;
#pragma unroll
        for (int t = 0; t < kQB; t++) {
            ty = throat_eval<float>(C, thr, Z.thY);
            if (t == 0) Z.thY = q_take<1, kPart1>(Z.thY, ty);
            if (t == 1) Z.thY = q_take<1, kPart2>(Z.thY, ty);
            if (t == 2) Z.thY = q_take<1, kPart3>(Z.thY, ty);
            if (t == 3) Z.thNext = q_take<1, kPart0>(Z.thNext, ty);
        }
        Z.thY = q_take<0, kPart0>(Z.thY, Z.thNext);
        if (saving && m + 1u == ntubeLane) st[2] = ty;
        return ty;
    };

#ifdef TRM_ABL_SKIP      // timing experiments only (tools/bench_variants.sh): the masked roles keep the barriers and do nothing
    if ((TRM_ABL_SKIP >> role) & 1) {
        for (uint32_t step = 0; step < nSteps; step++) step_barrier();
        return;
    }
#endif
    if (role == 0) {
#ifdef TRM_QPRIO_OSC
        __builtin_amdgcn_s_setprio(TRM_QPRIO_OSC);
#endif
        // ------------------------------------------------------------ osc: block i at step i, lane = (voice, slot)
        auto sine = [&](int i) { return sine_table(i); };
        OscSlotTrack T;
        double P = 0.0;                                 // oscillator position at the start of the block
        if (streaming && !sFirst) P = *reinterpret_cast<const double *>(st);
        if (kSeg) {
            // the oscillator's position at the warm-up start: the advances between the warm-up starts of the segments so far
            // (trm_phase_segment_kernel), summed and wrapped in order -- exact
            const double *ph = A.seg_phase + vRaw;
            const size_t pitch = (size_t)A.seg_wg_per_seg * kQV;
            for (uint32_t q = 1; q <= seg; q++) P = osc_wrap(P + ph[q * pitch]);
        }
        float prev[4], cur[4], nxt[4];                  // (streaming instance: the frames carried in registers)
        uint32_t per = 0, j = (uint32_t)part;           // control period / position in it of this lane's sample
        if (nSteps > 0) {
            if constexpr (kLdsFrames) {
                float fa[4], fb[4];
                frame_to(0, fa, 1);
                frame_to(1, fb, 1);
                osc_slot_setup(T, C, fa, fb, (int)j);
            } else {
                load_frame(frames, frame_index(0), prev, 1);
                load_frame(frames, frame_index(1), cur, 1);
                load_frame(frames, frame_index(2), nxt, 1);
                if (sHold) for (int q = 0; q < 4; q++) prev[q] = cur[q];
                osc_slot_setup(T, C, prev, cur, (int)j);
            }
        }
        float2 *const ring = &sO[vq * kOStride];
        ScanState Z;
        scans_restore(Z, false, TRM_THROAT_IN_OSC);
        STAMP_DECL
        for (uint32_t step = 0; step < nSteps; step++) {
            STAMP_BEGIN
#pragma unroll
            for (int u = 0; u < kSub; u++) {
                // the blocks of step i-2: the mix wave's {sig, thr} were written during step i-1; the tube wave reads
                // the result from step i+1 on
                const uint32_t blk = (step - 2) * kSub + u;
                if (TRM_THROAT_IN_OSC && step >= 2 && blk * kQB < nTotal) {
                    float *const xr = reinterpret_cast<float *>(&sX[((blk % kXBufs) * kQB + part) * kXPitch + vq]);
                    xr[2] = throat_scan(Z, xr[2], blk * kQB + (uint32_t)part);
                }
            }
#pragma unroll
            for (int u = 0; u < kSub; u++) {
              const uint32_t oblk = step * kSub + u;
              if (oblk * kQB < nTotal) {
                if (j >= CP) {      // this lane's sample starts a control period (:289); the next frame was prefetched
                    j -= CP;
                    per++;
                    if constexpr (kLdsFrames) {
                        float fa[4], fb[4];
                        frame_to(per, fa, 1);
                        frame_to(per + 1, fb, 1);
                        osc_slot_setup(T, C, fa, fb, (int)j);
                    } else {
                        for (int q = 0; q < 4; q++) { prev[q] = sHold ? nxt[q] : cur[q]; cur[q] = nxt[q]; }
                        load_frame(frames, frame_index(per + 2), nxt, 1);
                        osc_slot_setup(T, C, prev, cur, (int)j);
                    }
                }
                const double db = __builtin_fma((double)j, T.glotDelta, T.glot0);
                double axd = db >= 60.0 ? 1.0 : T.axGeo;      // amplitude() with its clamps (:294-296)
                axd = db <= 0.0 ? 0.0 : axd;
                const float ah1 = amplitude_f(fma_f((float)j, T.aspDelta, T.aspBase));
                const double oinc = osc_increment(T.f0, C);       // (a multiple of 2^-30: the sums below are exact)
                // position after this lane's sample = P + the inclusive prefix sum of 2*inc over the slots
                double pre = oinc + oinc;
                pre += q_take<1, kPart1 | kPart2 | kPart3>(0.0, pre);
                pre += q_take<2, kPart2 | kPart3>(0.0, pre);
                const double end = P + pre;
                const double pos2 = osc_wrap(end), pos1 = osc_wrap(end - oinc);
                if (saving && oblk * kQB + (uint32_t)part + 1u == ntubeLane) *reinterpret_cast<double *>(st) = pos2;   // the chunk's last sample
                double tot = pre;                                // slot 3's prefix = the block's advance
                tot = q_take<1, kPart0>(tot, pre);
                tot = q_take<2, kPart1>(tot, pre);
                tot = q_take<3, kPart2>(tot, pre);
                P = osc_wrap(P + tot);
                float wa, wb;
                osc_read(C, axd, pos1, pos2, sine, wa, wb);
                T.f0 *= T.f0Step;
                T.axGeo *= T.axStep;
                j += kQB;
                const uint32_t slot = (oblk * kQB + (uint32_t)part) & (kORing - 1);
                ring[slot] = make_float2(wa, wb);
                if (slot < (uint32_t)kOMirror) ring[slot + kORing] = make_float2(wa, wb);
                sA[(oblk % kABufs) * kWave + lane] = make_float2((float)axd, ah1);
              }
            }
            STAMP_MID
            step_barrier();
            STAMP_END
        }
        STAMP_STORE(role)
        if (saving && laneValid) {
            // the chunk's last 32 oscillator reads (positions N-32 .. N-1; older than the chunk: still in the ring)
            for (int t = part; t < 32; t += 4) {
                const float2 x = ring[(ntubeLane - 32u + (uint32_t)t) & (kORing - 1)];
                st[8 + 2 * t] = x.x;
                st[9 + 2 * t] = x.y;
            }
        }
    } else if (role == 1) {
#ifdef TRM_QPRIO_MIX
        __builtin_amdgcn_s_setprio(TRM_QPRIO_MIX);
#endif
        // ------------------------------------------------------------ mix: block i-1 at step i, lane = (voice, slot)
        const float *const lpNoise = A.lp_noise + (kSeg ? nBase : 0u);      // (a stream's pointer arrives advanced)
        auto fill_noise_half = [&](uint32_t nFirst, int half) {
            dma4(lpNoise + nFirst + lane, &sNoise[half * kNoiseHalf]);
        };
        // window taps of this lane's parity (m & 1 == part & 1: blocks start on multiples of 4), as (a, b) pairs
        const int o = part & 1;
        v2f cab[kFirWin];
        for (int i = 0; i < kFirWin; i++)
            cab[i] = o ? v2f{fir_window_tap_a(C.fir, 1, i), fir_window_tap_b(C.fir, 1, i)}
                       : v2f{fir_window_tap_a(C.fir, 0, i), fir_window_tap_b(C.fir, 0, i)};
        if (nSteps > 0) {
            fill_noise_half(0, 0);
            fill_noise_half(kNoiseHalf, 1);
            dma_wait_all();
        }
        const float2 *const ring = &sO[vq * kOStride];
        // Converter coefficient rows, staged for the convert wave: block B's 32 rows (128 bytes each, shifted per
        // output like the wide kernel's fetch) are loaded when the oscillator's time is within 4 samples of the
        // block's first output, written to LDS one step later and visible one step after that -- normally several
        // steps before the convert wave can begin the block.  Two words in LDS make that independent of timing:
        // sRowSync[1] = blocks staged (the convert wave begins no block beyond it), sRowSync[0] = the first block
        // whose rows the convert wave has not copied yet (buffer B % 3 is not rewritten before).
        ScanState Z;
        scans_restore(Z, false, !TRM_THROAT_IN_OSC);
        uint32_t rowBlk = 0;
        bool rowsInFlight = false;
        float4 rq[4];
        const uint32_t cvtOutputs = kSeg ? wave_max_u32(noutSeg) : streaming ? A.stream_k_end - kBase
                                              : wave_max_u32(laneValid && nfr > 0 ? (uint32_t)((((uint64_t)ntubeLane + 2ull * (uint32_t)C.padSize) * 65536ull + inc - 1) / inc) : 0u);
        const uint32_t cvtBlocks = C.upsample ? (cvtOutputs + kCvtCols - 1) / kCvtCols : 0;
        STAMP_DECL
        for (uint32_t step = 0; step < nSteps; step++) {
            STAMP_BEGIN
            auto rows_to_lds = [&]() {
                float4 *dst = reinterpret_cast<float4 *>(&sRows[((rowBlk - 1) % kRowBufs) * (kCvtCols * kRowPitch) + (lane >> 1) * kRowPitch + (lane & 1) * 16]);
                for (int q = 0; q < 4; q++) dst[q] = rq[q];
                rowsInFlight = false;
                lds_flag_publish(&sRowSync[1], rowBlk, lane == 0);     // blocks 0 .. rowBlk-1 are in LDS
            };
            if (rowsInFlight) rows_to_lds();
            // never more than kRowBufs blocks past the first one the convert wave still has to copy (looked up only
            // when a block is due: about once per block).  Up to two blocks per step (rate ratios above 4).
            for (int r = 0; r < 2; r++) {
                if (TRM_ABL_CVT != 4 && rowBlk < cvtBlocks && src_position(kBase + rowBlk * kCvtCols, inc) - nBase <= step * kStepN + 4u &&
                    rowBlk < lds_flag_consume(&sRowSync[0]) + kRowBufs) {
                    if (rowsInFlight) rows_to_lds();            // (a second block in the same step: its predecessor's loads are waited for here)
                    const uint32_t k = kBase + rowBlk * kCvtCols + ((uint32_t)lane >> 1);
                    const uint32_t off = (src_position(k, inc) - nBase + (kQLead - (kSrcWindow - 1))) & 3u;
                    const float *pc = A.src_rows + (size_t)src_phase(k, inc) * kSrcRowC - off + (lane & 1) * 16;
                    for (int q = 0; q < 4; q++) rq[q] = make_float4(pc[4 * q], pc[4 * q + 1], pc[4 * q + 2], pc[4 * q + 3]);
                    rowsInFlight = true;
                    rowBlk++;
                }
            }
#pragma unroll
            for (int u = 0; u < kSub; u++) {
              const uint32_t blk = (step - 1) * kSub + u;
              if (step >= 1 && blk * kQB < nTotal) {
                const int buf = blk % kABufs, xbuf = blk % kXBufs;
                const uint32_t n0 = blk * kQB;
                if ((n0 & (kNoiseHalf - 1)) == 0 && n0 > 0) {
                    // entering a noise half: it was requested one half ago; refill the other half
                    dma_wait_all();
                    fill_noise_half(n0 + kNoiseHalf, ((n0 / kNoiseHalf) + 1) & 1);
                }
                const uint32_t m = n0 + (uint32_t)part;
                // 26-sample window starting at the even index m - 24 - o (zeros before the first sample)
                const uint32_t s0 = (m + kORing - 24u - (uint32_t)o) & (kORing - 1);
                const float4 *wp = reinterpret_cast<const float4 *>(&ring[s0]);
                const float pulse = fir_window_dot(wp, cab);
                const float2 a = sA[buf * kWave + lane];
                const Excitation E = mix_tail(C, a.x, a.y, pulse, sNoise[m & (kNoiseRing - 1)]);
#if TRM_THROAT_IN_OSC
                const float ty = E.thr;         // (raw: the oscillator wave turns it into the throat output two steps on)
#else
                const float ty = throat_scan(Z, E.thr, m);
#endif
                sX[(xbuf * kQB + part) * kXPitch + vq] = make_float4(E.gin, E.sig, ty, 0.0f);
              }
            }
            STAMP_MID
            step_barrier();
            STAMP_END
        }
        STAMP_STORE(role)
        dma_wait_all();
    } else if (role == 2 || role == 3) {
#ifdef TRM_QPRIO_COEF
        if (role == 2) __builtin_amdgcn_s_setprio(TRM_QPRIO_COEF);
#endif
#ifdef TRM_QPRIO_COEF2
        if (role == 3) __builtin_amdgcn_s_setprio(TRM_QPRIO_COEF2);
#endif
        // ------------------------------------------------------------ coef: block i-2 at step i, lane = (voice, slot).
        // Two waves share the stage by FUNCTION: role 2 turns radii and velum into the junctions' transmission factors
        // (stateless in time), role 3 turns the frication tracks into taps and band-pass coefficients, runs the band-pass
        // over the mix wave's noise signal (written during step i-1) and hands the tube the INJECTIONS tap x band-pass
        // output: the tube wave neither sees the band-pass nor multiplies by taps.
        const bool area = role == 2;
        ScanState Z;
        scans_restore(Z, !area, false);
        CoefTrack T;
        float prev[16], cur[16], nxt[16];               // (streaming instance only)
        uint32_t per = 0, j = (uint32_t)part;
        if (nSteps > 0) {
            if constexpr (kLdsFrames) {
                float fa[16], fb[16];
                frame_to(0, fa, 4);
                frame_to(1, fb, 4);
                coef_track_setup(T, C, fa, fb);
            } else {
                load_frame(frames, frame_index(0), prev, 4);
                load_frame(frames, frame_index(1), cur, 4);
                load_frame(frames, frame_index(2), nxt, 4);
                if (sHold) for (int q = 0; q < 16; q++) prev[q] = cur[q];
                coef_track_setup(T, C, prev, cur);
            }
        }
        // the stager (area wave of a one-shot instance): the oscillator wave, two steps ahead of this one, enters the
        // period between frames p and p + 1 in the step that holds sample p * CP
        uint32_t stP = 1, stBnd = CP, stSend = 0;
        STAMP_DECL
        for (uint32_t step = 0; step < nSteps; step++) {
            STAMP_BEGIN
            if (kLdsFrames && area) {
                if (stSend) { stage_frame(stSend); stSend = 0; }
                if (step * kStepN < nTotal && step * kStepN + (kStepN - 1) >= stBnd) {
                    dma_wait_all();              // frame stP + 2, sent for a period ago
                    stSend = stP + 3;            // its slot is frame stP - 1's: free from the next step on
                    stP++;
                    stBnd += CP;
                }
            }
#pragma unroll
            for (int u = 0; u < kSub; u++) {
              const uint32_t blk = (step - 2) * kSub + u;
              if (step >= 2 && blk * kQB < nTotal) {
                const int buf = blk % kKBufs;
                if (j >= CP) {
                    j -= CP;
                    per++;
                    if constexpr (kLdsFrames) {
                        float fa[16], fb[16];
                        frame_to(per, fa, 4);
                        frame_to(per + 1, fb, 4);
                        coef_track_setup(T, C, fa, fb);
                    } else {
                        for (int q = 0; q < 16; q++) { prev[q] = sHold ? nxt[q] : cur[q]; cur[q] = nxt[q]; }
                        load_frame(frames, frame_index(per + 2), nxt, 4);
                        coef_track_setup(T, C, prev, cur);
                    }
                }
                Coefs K;
                PartRecord R[4];
                SharedRecord H;
                // [buf][slot]{kk | tp}[the tube wave's lane of (voice, part p)]
                float4 *dst = &sK[(buf * kQB + part) * kKPitch + (vq >> 2) * 16 + (vq & 3)];
                if (area) {
                    coef_sample_area(K, T, C, (int)j);
                    pack_part_kk(K, C, R);
                    pack_shared_end(K, C, H);
                    for (int p = 0; p < 4; p++) dst[p * 4] = make_float4(R[p].kk[0], R[p].kk[1], R[p].kk[2], R[p].kk[3]);
                    sBP[(buf * kQB + part) * kXPitch + vq] = make_float4(H.endK[0], H.endK[1], H.endOnePlus[0], H.endOnePlus[1]);
                } else {
                    coef_sample_fric<kStream>(K, T, C, (int)j);
                    pack_part_tp(K, R);
                    pack_shared_bp(K, H);
                    const float f = bandpass_scan(Z, blk, make_float4(H.bpA2, H.bpB2, H.bpG2, 0.0f));
                    for (int p = 0; p < 4; p++) dst[kWave + p * 4] = make_float4(R[p].tp[0] * f, R[p].tp[1] * f, R[p].tp[2] * f, R[p].tp[3] * f);
                }
                j += kQB;
              }
            }
            STAMP_MID
            step_barrier();
            STAMP_END
        }
        STAMP_STORE(role)
        if (kLdsFrames && area) dma_wait_all();     // nothing may still be writing LDS when the wave ends
    } else if (role == 4) {
        // ------------------------------------------------------------ tube: block i-4 at step i, lane = (voice, part)
        // the only serial role: where it shares a SIMD (two workgroups on a CU) its instructions go first (8192 voices:
        // 4.54 -> 4.33 ms, profiles/role_perm_r02.txt)
#ifndef TRM_QPRIO_TUBE
#define TRM_QPRIO_TUBE 3
#endif
        __builtin_amdgcn_s_setprio(TRM_QPRIO_TUBE);
        QuadState<float> S;
        quad_reset(S);
        float *const stTube = saving ? st + 104 + 20 * part : nullptr;
        if (streaming && !sFirst) {
            S.TA = v2f_t{stTube[0], stTube[1]}; S.TB = v2f_t{stTube[2], stTube[3]};
            S.BA = v2f_t{stTube[4], stTube[5]}; S.BB = v2f_t{stTube[6], stTube[7]};
            S.A0 = stTube[8]; S.jT = stTube[9]; S.jB = stTube[10]; S.jN = stTube[11];
            S.eB = v2f_t{stTube[12], stTube[13]}; S.reflY = v2f_t{stTube[14], stTube[15]};
            S.radX = v2f_t{stTube[16], stTube[17]}; S.radY = v2f_t{stTube[18], stTube[19]};
        }
        auto save_state = [&]() {
            stTube[0] = S.TA.x; stTube[1] = S.TA.y; stTube[2] = S.TB.x; stTube[3] = S.TB.y;
            stTube[4] = S.BA.x; stTube[5] = S.BA.y; stTube[6] = S.BB.x; stTube[7] = S.BB.y;
            stTube[8] = S.A0; stTube[9] = S.jT; stTube[10] = S.jB; stTube[11] = S.jN;
            stTube[12] = S.eB.x; stTube[13] = S.eB.y; stTube[14] = S.reflY.x; stTube[15] = S.reflY.y;
            stTube[16] = S.radX.x; stTube[17] = S.radX.y; stTube[18] = S.radY.x; stTube[19] = S.radY.y;
        };
        float4 *const ring = reinterpret_cast<float4 *>(&sY[vq * kYStride]);
        float *const tubeOut = A.tube_out ? A.tube_out + A.tube_offset[v] : nullptr;
        // one sample's inputs: {gin, -, throat output}, end coefficients, this part's record {transmission | injection}
        struct In { float4 x, e4, k4, t4; };
        auto load_in = [&](uint32_t blk, int s) {
            const int buf = blk % kKBufs;
            In r;
            r.x = sX[((blk % kXBufs) * kQB + s) * kXPitch + vq];
            r.e4 = sBP[(buf * kQB + s) * kXPitch + vq];
            const float4 *rec = &sK[(buf * kQB + s) * kKPitch + lane];
            r.k4 = rec[0];
            r.t4 = rec[kWave];
            return r;
        };
        auto step_one = [&](const In &r) {
            return tube_quad_core<float>(S, C, r.x.x, r.x.z, v2f_t{r.e4.x, r.e4.y}, v2f_t{r.e4.z, r.e4.w},
                                         v2f_t{r.k4.x, r.k4.y}, v2f_t{r.k4.z, r.k4.w}, v2f_t{r.t4.x, r.t4.y},
                                         v2f_t{r.t4.z, r.t4.w});
        };
        // The blocks of step i-4 at step i (their coefficient records were written during step i-2).  A block's first
        // sample's inputs are fetched behind the last sample of the block before it (for the step's first block:
        // during step i-1), the other three land behind the first sample's arithmetic: no LDS latency is exposed.
        In head;
        head.x = head.e4 = head.k4 = head.t4 = make_float4(0.f, 0.f, 0.f, 0.f);
        STAMP_DECL
        for (uint32_t step = 0; step < nSteps; step++) {
            STAMP_BEGIN
#pragma unroll
            for (int u = 0; u < kSub; u++) {
              const uint32_t blk = (step - 4) * kSub + u;
              if (step >= 4 && blk * kQB < nTotal) {
                const uint32_t n0 = blk * kQB;
                const In i1 = load_in(blk, 1), i2 = load_in(blk, 2), i3 = load_in(blk, 3);
                float y[kQB];
                // (samples past nTotal in the last block step on stale inputs; their output is forced to 0)
                // (streaming: the state after the chunk's last sample is what the next chunk starts from)
                const bool saveHere = saving && n0 < ntubeLane && n0 + kQB >= ntubeLane;      // uniform
                y[0] = step_one(head);
                if (saveHere && n0 + 1 == ntubeLane) save_state();
                y[1] = step_one(i1);
                if (saveHere && n0 + 2 == ntubeLane) save_state();
                y[2] = step_one(i2);
                if (saveHere && n0 + 3 == ntubeLane) save_state();
                head = load_in(blk + 1, 0);
                y[3] = step_one(i3);
                if (saveHere && n0 + 4 == ntubeLane) save_state();
                if (n0 + kQB > ntubeMin) {     // (uniform) zero flush / voices shorter than the group's longest
#pragma unroll
                    for (int s = 0; s < kQB; s++) y[s] = n0 + s < ntubeLane ? y[s] : 0.0f;
                }
                if (part == 2) {
                    // tube sample n sits at ring slot (n + kQLead) & 127: a block is one aligned 16-byte store
                    const uint32_t slot4 = ((n0 + kQLead) & (kYRing - 1)) >> 2;
                    const float4 yy = make_float4(y[0], y[1], y[2], y[3]);
                    ring[slot4] = yy;
                    if (slot4 < (uint32_t)(kYMirror / 4)) ring[slot4 + kYRing / 4] = yy;
                    if (tubeOut && laneValid) {
                        // (rows of tube_out are 16-byte aligned: the host pads their pitch to 4 floats)
                        const uint32_t lim = ntubeLane + (sLast ? 2u * (uint32_t)C.padSize : 0u);
                        if (n0 + kQB <= lim) *reinterpret_cast<float4 *>(tubeOut + n0) = yy;
                        else
                            for (int s = 0; s < kQB; s++)
                                if (n0 + s < lim) tubeOut[n0 + s] = y[s];
                    }
                }
              }
            }
            if (step == 3 && nTotal > 0) head = load_in(0, 0);
            STAMP_MID
            step_barrier();
            STAMP_END
        }
        STAMP_STORE(role)
    } else {
#ifdef TRM_QPRIO_CVT
        __builtin_amdgcn_s_setprio(TRM_QPRIO_CVT);
#endif
        // ------------------------------------------------------------ convert (lane = output time), 16 voices
        uint32_t noutLane = 0;
        if (kSeg) {
            noutLane = noutSeg;
        } else if (streaming) {
            noutLane = A.stream_k_end - kBase;
        } else if (nfr > 0) {
            uint64_t total = (uint64_t)ntubeLane + 2ull * (uint32_t)C.padSize;
            noutLane = (uint32_t)((total * 65536ull + inc - 1) / inc);
        }
        if (!laneValid) noutLane = 0;
        const uintptr_t myOut = reinterpret_cast<uintptr_t>(A.out + A.out_offset[v] + (kSeg ? kBase : 0u));
        const uint32_t noutMax = wave_max_u32(noutLane);
        const uint32_t nBlocks = C.upsample ? (noutMax + kCvtCols - 1) / kCvtCols : 0;
        const int col = lane & (kCvtCols - 1);
        const bool upper = lane >= kCvtCols;
        if (part == 0) sInfo[vq] = make_uint4(noutLane, (uint32_t)myOut, (uint32_t)(myOut >> 32), 0u);
        // running max |y| per (row, lane): row r = voices 2r (lanes 0-31) and 2r+1 (lanes 32-63)
        for (int r = 0; r < 8; r++) sMx[r * kWave + lane] = 0.0f;

        // output k reads tube samples e-25 .. e, e = src_position(k): ring slots e + kRingShift .. + 25
        constexpr uint32_t kRingShift = kQLead - (kSrcWindow - 1);
        // Coefficient rows: the mix wave stages every block's 32 rows in LDS several steps ahead (the latency of
        // global memory would otherwise sit in front of each block's first row pair: this wave's vector-memory
        // counter also counts its PCM stores); a block begins by copying this lane's row into registers.
        v2f cc[16];
        uint32_t blk = 0, pr = 0;       // next work item: row pair `pr` (0..3) of block `blk`: voices 4*pr .. 4*pr+3
        uint32_t winBase = 0, kLane = 0, needLast = 0, needNext = 0;
        // kLane = this lane's output within the launch; its global index is kBase + kLane, and ring positions are
        // relative to the launch's first tube sample (nBase)
        auto begin_block = [&]() {
            kLane = blk * kCvtCols + col;
            winBase = (src_position(kBase + kLane, inc) - nBase + kRingShift) & (kYRing - 1) & ~3u;
            needLast = src_position(kBase + blk * kCvtCols + (kCvtCols - 1), inc) - nBase;
            needLast = needLast < nTotal - 1 ? needLast : nTotal - 1;
            needNext = src_position(kBase + (blk + 1) * kCvtCols + (kCvtCols - 1), inc) - nBase;   // (past the end: never "behind")
            const float4 *row = reinterpret_cast<const float4 *>(&sRows[(blk % kRowBufs) * (kCvtCols * kRowPitch) + col * kRowPitch]);
            for (int q = 0; q < 8; q++) {
                const float4 x = row[q];
                cc[2 * q] = v2f{x.x, x.y};
                cc[2 * q + 1] = v2f{x.z, x.w};
            }
        };
        auto begin_block_from_global = [&]() {
            kLane = blk * kCvtCols + col;
            winBase = (src_position(kBase + kLane, inc) - nBase + kRingShift) & (kYRing - 1) & ~3u;
            needLast = nTotal - 1;
            const uint32_t off = (src_position(kBase + kLane, inc) - nBase + kRingShift) & 3u;
            const float *pc = A.src_rows + (size_t)src_phase(kBase + kLane, inc) * kSrcRowC - off;
            for (int q = 0; q < 16; q++) cc[q] = v2f{pc[2 * q], pc[2 * q + 1]};
        };
        bool needBegin = nBlocks > 0;
        typedef __attribute__((address_space(1))) float *GlobalFloatPtr;
        typedef __attribute__((address_space(3))) float *LdsFloatPtr;
        // metering (16.16 row pairs per step): a step's kQB tube samples turn into kQB * 2^16/inc outputs per
        // voice = that / 32 blocks of 4 row pairs
        const uint32_t earn = (uint32_t)(((uint64_t)kStepN << 32) / inc / 8) + 2048;
        // the cap must leave room to catch up after waiting for a block (a cap of about `earn` loses credit while
        // it waits and the wave falls behind until the ring laps it: seen at a 30 cm tube, ratio 3.8)
        const uint32_t capPairs = (earn + 0x18000u) >> 16;                  // floor(earn + 1.5)
        const uint32_t creditCap = (capPairs > 2u ? capPairs : 2u) << 16;   // two pairs per step at speech rates (earn ~ 1.1)
        uint32_t credit = 0;
        auto do_pair = [&]() {
#if TRM_ABL_CVT == 3 || TRM_ABL_CVT == 4     /* (timing experiments: 3 = the converter's control flow only; 4 = and no row staging) */
            if (++pr == 4) { pr = 0; blk++; needBegin = blk < nBlocks; }
            return;
#endif
            const int la = 4 * (int)pr, lb = la + 2;
            const int ha = upper ? 1 : 0;
            const float4 *wa = reinterpret_cast<const float4 *>(&sY[(la + ha) * kYStride + winBase]);
            const float4 *wb = reinterpret_cast<const float4 *>(&sY[(lb + ha) * kYStride + winBase]);
            const uint4 ia = sInfo[la + ha], ib = sInfo[lb + ha];
            // (the second row's window is read while the first row's chains run: one row's registers are live at a time)
            const float ya = cvt_window_dot(wa, cc), yb = cvt_window_dot(wb, cc);
            const bool okA = kLane < ia.x, okB = kLane < ib.x;
#if TRM_ABL_CVT != 1      /* (timing experiments: 1 = no PCM stores) */
            if (okA) reinterpret_cast<GlobalFloatPtr>(((uintptr_t)ia.z << 32) | ia.y)[kLane] = ya;
            if (okB) reinterpret_cast<GlobalFloatPtr>(((uintptr_t)ib.z << 32) | ib.y)[kLane] = yb;
#endif
            __builtin_amdgcn_ds_fmaxf((LdsFloatPtr)&sMx[(2 * pr) * kWave + lane], okA ? fabsf(ya) : 0.0f, 0, 0, false);
            __builtin_amdgcn_ds_fmaxf((LdsFloatPtr)&sMx[(2 * pr + 1) * kWave + lane], okB ? fabsf(yb) : 0.0f, 0, 0, false);
            if (++pr == 4) {
                pr = 0;
                blk++;
                needBegin = blk < nBlocks;
            }
        };
        STAMP_DECL
        for (uint32_t step = 0; step < nSteps; step++) {
            STAMP_BEGIN
            // visible after the previous barrier: tube samples n < (step-4)*kStepN
            const uint32_t ready = step >= 4 ? (step - 4) * kStepN : 0;
            credit += earn;
            if (credit > creditCap) credit = creditCap;     // a ready block is spread over the next steps, not done in a burst
            // (the two sync words are touched once per block: when a block has to begin)
            auto try_begin = [&]() {
                if (TRM_ABL_CVT == 4 || blk < lds_flag_consume(&sRowSync[1])) {
                    begin_block();
                    needBegin = false;
                    lds_flag_publish(&sRowSync[0], blk + 1, lane == 0);    // this block's rows are in registers now
                }
            };
            if (needBegin) try_begin();
            // Metered by the credit (smooth: at most creditCap pairs in a step) -- but a wave that is a whole block
            // behind production works through its backlog regardless: the credit alone loses what it earns while it
            // waits for a block to complete, and at some rate ratios that slowly let the ring lap the converter.
            while (blk < nBlocks && !needBegin && needLast < ready && (credit >= (1u << 16) || needNext < ready)) {
                credit = credit >= (1u << 16) ? credit - (1u << 16) : credit;
                do_pair();
                if (needBegin) try_begin();
            }
            STAMP_MID
            step_barrier();
            STAMP_END
        }
        STAMP_STORE(role)
        {   // after the last barrier: what is staged is final; a block beyond it fetches its row itself
            const uint32_t staged = lds_flag_consume(&sRowSync[1]);
            while (blk < nBlocks) {
                if (needBegin) {
                    if (blk < staged) begin_block();
                    else begin_block_from_global();
                    needBegin = false;
                }
                do_pair();
            }
        }
        if (saving && laneValid && part < 2) {
            // the chunk's last 32 tube samples (positions N-32 .. N-1) for the next chunk's converter
            for (int t = part; t < 32; t += 2)
                st[72 + t] = sY[vq * kYStride + ((ntubeLane - 32u + (uint32_t)t + kQLead) & (kYRing - 1))];
        }
        float myMax = 0.0f;     // collected by lanes 0..15: voice `lane` of the workgroup
#pragma unroll
        for (int r = 0; r < 8; r++) {
            float m = sMx[r * kWave + lane];
            for (int off = 16; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, kWave));
            const float lowHalf = __shfl(m, 0, kWave), highHalf = __shfl(m, 32, kWave);
            if (lane == 2 * r) myMax = lowHalf;
            if (lane == 2 * r + 1) myMax = highHalf;
        }
        const uint32_t ov = vblock * kQV + (uint32_t)lane;
        if (lane < kQV && ov < A.nvoices && C.upsample) {
            const uint32_t nf = min(A.nframes[ov], A.max_nframes);
            uint32_t nov = 0;
            if (streaming && !kSeg) nov = A.stream_k_end - kBase;
            else if (nf > 0) nov = (uint32_t)((((uint64_t)(nf - 1) * CP + 2ull * (uint32_t)C.padSize) * 65536ull + inc - 1) / inc);
            if (kSeg) {
                // (max_sample was zeroed by the launcher; non-negative floats order like their bit patterns)
                if (seg == 0) A.number_samples[ov] = nov;
                if (myMax > 0.0f) atomicMax(reinterpret_cast<unsigned int *>(&A.max_sample[ov]), __float_as_uint(myMax));
            } else {
                A.number_samples[ov] = nov;
                A.max_sample[ov] = myMax;
            }
        }
        return;
    }
}

// `cus`: the device's compute units.  A batch of more workgroups than that runs the instance that fits two of them on
// a CU (kSub = 1); up to one workgroup per CU the instance with two independent blocks per step (kSub = 2).
template <bool kStream, int kSub, bool kSeg = false>
static hipError_t launch_instance(const Const &c, const TubeArgs &a, hipStream_t stream, uint32_t grid)
{
    static DynamicLdsAllowance lds;
    hipError_t e = lds.ensure(reinterpret_cast<const void *>(trm_tube_kernel_q<kStream, kSub, kSeg>), (int)QuadLds<kSub>::kBytes);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((trm_tube_kernel_q<kStream, kSub, kSeg>), dim3(grid), dim3(kWave * kQRoles), QuadLds<kSub>::kBytes, stream, c, a);
    return hipGetLastError();
}

hipError_t launch_tube_quad(const Const &c, const TubeArgs &a, hipStream_t stream, int cus)
{
    if (a.nvoices == 0) return hipSuccess;
    uint32_t grid = (a.nvoices + kQV - 1) / kQV;
    if (a.seg_periods) return launch_instance<true, 2, true>(c, a, stream, a.seg_grid);      // time split: 16 voices x one segment per workgroup
    if (a.stream_state) return launch_instance<true, 2>(c, a, stream, grid);
    // one-shot instances stage the control frames in a ring of four: frame p+3 replaces frame p-1 one step into period p,
    // and the coefficient waves' last lanes read frame p-1 three steps into period p-1 -- a period must hold three steps
    // of up to 8 samples (the caller runs trm_kernels.hip's kernel otherwise)
    if (c.controlPeriod < 24) return hipErrorInvalidValue;
    if (cus > 0 && grid > (uint32_t)cus) return launch_instance<false, 1>(c, a, stream, grid);
    return launch_instance<false, 2>(c, a, stream, grid);
}

int tube_quad_kernel_blocks_per_cu(int sub)
{
    int n = 0;
    hipError_t e = sub == 1 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, trm_tube_kernel_q<false, 1>, kWave * kQRoles, QuadLds<1>::kBytes)
                            : hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, trm_tube_kernel_q<false, 2>, kWave * kQRoles, QuadLds<2>::kBytes);
    return e == hipSuccess ? n : -1;
}

}  // namespace trm
