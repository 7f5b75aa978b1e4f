// trm_kernels.hip -- CDNA4 (gfx950) kernels of the Tube Resonance Model.
//
//   trm_noise_kernel      the voice-independent noise sequence (TRMUtility.m:71-85 + TRMFilters.m:81-86),
//                         fp64 serial recurrence, one lane, run once per batch object and cached
//   trm_tube_kernel       -[TRMTubeModel synthesize] (TRMTubeModel.m:272-361), one tube per lane: 64 voices
//                         per workgroup, 7 waves per workgroup running the sample loop as a pipeline
//                         (osc | mix | coef x2 | tube | convert x2) with hand-offs through LDS; state in
//                         VGPRs; wave-uniform control in SGPRs; noise prefetched into an LDS ring by LDS-DMA;
//                         PCM written as rows straight from registers.  The form for batches that fill the
//                         chip; smaller ones run trm_quad.hip's four-lanes-per-voice form.
//   trm_downsample_kernel the converter's down-sampling branch (TRMSampleRateConverter.m:234-297)
//   trm_int16_kernel      output normalisation (TRMTubeModel.m:370-389, 420-484)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include <type_traits>

#include "trm_devutil.h"
#include "trm_kernels.h"
#include "trm_lane.h"

// Timing experiments live behind ONE switch (-DTRM_EXPERIMENTS, tools/build_variant.sh); the product build defines none of them.
#ifndef TRM_EXPERIMENTS
#undef TRM_ABL
#undef TRM_ROLE_PERM
#undef TRM_PRIO_OSC
#undef TRM_PRIO_MIX
#undef TRM_PRIO_COEF
#undef TRM_PRIO_TUBE
#undef TRM_PRIO_CVT
#undef TRM_PRIO_CVT_S
#undef TRM_PRIO_TUBE_S
#undef TRM_PRIO_OSC_S
#undef TRM_EXP_DOWN
#endif
#ifndef TRM_ABL
#define TRM_ABL 0    // diagnostic ablations of the convert stage; 0 in the product
#endif
// Issue priority per role (s_setprio, 0..3).  The kernel is VALU-throughput bound (91 % of the SIMDs' time is VALU issue with all
// roles at priority 0); what is left is waves waiting at the step barrier for the role that got its issue slots last.  Serving
// the roles in the order convert > tube > oscillator > {coefficients, mix} took the saturating batch from 17.2 to 15.5 ms
// (profiles/ab_r03.txt: 25 orders timed; the order is strict -- coefficient or mix waves above the oscillator cost 20-40 %).
// The streaming instance wants tube == oscillator: with the one-shot order a 100 ms chunk of 1 M voices takes 39 ms, with
// convert 3 > tube 1 = oscillator 1 it takes 27.7 ms (30.8 without priorities) -- TRM_PRIO_*_S, same file.
#ifndef TRM_PRIO_CVT
#define TRM_PRIO_CVT 3
#endif
#ifndef TRM_PRIO_TUBE
#define TRM_PRIO_TUBE 2
#endif
#ifndef TRM_PRIO_OSC
#define TRM_PRIO_OSC 1
#endif
#ifndef TRM_PRIO_CVT_S
#define TRM_PRIO_CVT_S 3
#endif
#ifndef TRM_PRIO_TUBE_S
#define TRM_PRIO_TUBE_S 1
#endif
#ifndef TRM_PRIO_OSC_S
#define TRM_PRIO_OSC_S 1
#endif
#ifndef TRM_PRIO_MIX
#define TRM_PRIO_MIX 0
#endif
#ifndef TRM_PRIO_COEF
#define TRM_PRIO_COEF 0
#endif

namespace trm {

constexpr int kRoles = 7;            // waves per workgroup: osc, mix, coef x2, tube, convert x2
constexpr int kTB = 2;               // tube samples per pipeline step (one barrier per step)
constexpr int kKQuads = 5;           // coef -> tube: 20 floats per lane per sample

__global__ void trm_noise_kernel(float *lp, uint32_t from, uint32_t to, double *state)
{
    // The generator is chaotic: the product must be rounded to double before the subtraction, exactly
    // as the reference does it (no fused multiply-add), or the sequence diverges within a few samples.
#pragma clang fp contract(off)
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    double seed = state[0], x1 = state[1];
    for (uint32_t i = from; i < to; i++) {
        double prod = seed * 377.0;
        asm volatile("" : "+v"(prod));       // opaque: keeps the rounded product, forbids v_fma_f64 fusion
        seed = prod - (double)(int)prod;
        double nz = seed - 0.5;
        lp[i] = (float)(nz + x1);
        x1 = nz;
    }
    state[0] = seed;
    state[1] = x1;
}

// One workgroup = 64 voices x 7 waves.
//   osc, mix, coef x2, tube: lane = voice.  At step i the osc wave produces block i (kTB tube samples), the
//     mix wave and the two coef waves (one per sample parity; the coefficient stage is stateless in time)
//     block i-1, the tube wave consumes block i-2 and appends its tube-rate samples to a per-voice ring
//     in LDS.  Hand-off buffers are double-buffered; one barrier per step.
//   convert x2: lane = OUTPUT TIME.  The converter is a feed-forward FIR, so it runs transposed: a
//     wave-row is 32 consecutive output samples of two voices (lanes 0-31 / 32-63).  All 64 voices
//     share the block's 32 phases, so each lane keeps its 26 coefficients in VGPRs for 32 rows; the
//     26-sample windows come out of the ring as conflict-free LDS reads, and every row is stored
//     straight from registers as two contiguous 128-byte pieces of PCM -- no transposition tile, no
//     per-output control flow.  The two convert waves take alternate row pairs; a fixed number of
//     rows per step keeps them ahead of production.
// kStream: the launch is one chunk of a streamed utterance (trm_kernels.h: state restored at its start, saved at its last
// tube sample; converter outputs keep their global index, ring positions follow the global tube-sample index); a
// compile-time switch so that the one-shot instance carries none of it.  Per voice kStreamFloats floats of state:
//   [0..1] oscillator position (fp64)   [2..25] the FIR's partial sums   [26..57] the 32 travelling waves
//   [58..68] filter memories            [72..103] the last 32 tube samples (the converter's history)
// stored per workgroup as [field][lane]: field i of the workgroup's lane l at stream_state[(wg * kStreamFloats + i) * 64 + l]
// (trm_quad.hip's records are voice-major); the buffer holds kStreamFloats floats per voice, voices rounded up to 64.
// kMode: kModeOneShot | kModeStream (above) | kModeSegments: a time-split launch (trm_kernels.h, TubeArgs::seg_*): the workgroup
// runs ONE segment of its 64 voices from rest, a warm-up ahead of the segment's first control period; like a stream chunk
// its tube samples and converter outputs keep their global indices (nBase, kBase -- here per workgroup), unlike one it
// neither restores nor saves state: only the oscillator position is handed in.
constexpr int kModeOneShot = 0, kModeStream = 1, kModeSegments = 2;
template <int kMode>
__global__ __launch_bounds__(kWave *kRoles, 4) void trm_tube_kernel(const Const C, const TubeArgs A)
{
    constexpr bool kStream = kMode == kModeStream, kSeg = kMode == kModeSegments;
    // (two launches of one batch, one of which runs: TubeArgs::gate)
    if (A.gate && ((*A.gate != 0u) ? 1u : 0u) != A.gate_want) return;
    __shared__ __attribute__((aligned(16))) float4 sW[2 * kTB * kWave];          // osc -> mix: {wa, wb, ax, ah1}
    __shared__ __attribute__((aligned(16))) float4 sX[2 * kTB * kWave];          // mix -> tube: excitation per sample
    __shared__ __attribute__((aligned(16))) float4 sK[2 * kTB * kKQuads * kWave]; // coefficients per sample
    __shared__ __attribute__((aligned(16))) float sY[kWave * kYStride];          // tube-rate rings
    __shared__ uint4 sInfo[kWave];                                               // convert: {length, ptr lo, ptr hi} per voice
    __shared__ float sMx[2 * 16 * kWave];                                        // convert: running max |y| per (row, lane)
    __shared__ float sNoise[kNoiseRing];                                         // excite: noise ring

    constexpr int kStampRoles = kRoles;
    (void)kStampRoles;
    const int lane = threadIdx.x & (kWave - 1);
    // wave -> role.  A workgroup's waves are dealt to the CU's 4 SIMDs in turn, so waves w and w+4 share
    // one SIMD's issue slots: TRM_ROLE_PERM lists the role of each wave (diagnostic builds may override it).
#ifndef TRM_ROLE_PERM
#define TRM_ROLE_PERM 5, 6, 4, 0, 2, 3, 1   /* convert0 convert1 tube osc | coef0 coef1 mix (140 of the 1260 orders timed, profiles/ab_r03.txt) */
#endif
    const int waveIdx = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int rolePerm[kRoles] = {TRM_ROLE_PERM};
    int role = 0;
    for (int i = 0; i < kRoles; i++) role = waveIdx == i ? rolePerm[i] : role;
    const uint32_t wg = A.wg_base + blockIdx.x;          // (a large batch is launched in slices: launch_tube)
    // time-split: workgroup -> (segment, block of 64 voices)
    uint32_t seg = 0, vblock = wg;
    if (kSeg) {
        if (A.seg_map) {                     // (the pairs with work first: trm_seg_map_kernel)
            const uint2 m = A.seg_map[wg];
            seg = m.x;
            vblock = m.y;
        } else {
            seg = wg / A.seg_wg_per_seg;
            vblock = wg - seg * A.seg_wg_per_seg;
        }
    }
    const uint32_t vRaw = vblock * kWave + lane;
    const bool laneValid = vRaw < A.nvoices;
    const uint32_t v = laneValid ? vRaw : A.nvoices - 1;
    const uint32_t CP = (uint32_t)C.controlPeriod;
    const uint32_t inc = C.timeRegisterIncrement;
    // converter outputs with a read position before tube sample `end`: k < outputs_before(end) (trm_capi.cc outputs_through)
    auto outputs_before = [&](uint64_t end) { return end == 0 ? 0u : (uint32_t)(((end << 16) - 1) / inc + 1); };

    // segment s covers the control periods seg_begin(s) .. seg_begin(s + 1): the first segment is a warm-up longer than the
    // others (it has none of its own), so that every workgroup of the launch runs the same number of periods
    auto seg_begin = [&](uint32_t sgm) { return sgm == 0 ? 0u : A.seg_first + (sgm - 1) * A.seg_periods; };
    const uint32_t nfrAll = min(A.nframes[v], A.max_nframes);
    // the frames this launch runs for this lane: the utterance's, or those of the workgroup's segment with its warm-up
    uint32_t nfr = nfrAll, segFrame0 = 0, segOutEnd = 0;
    bool segLast = true;
    if (kSeg) {
        const uint32_t nper = nfrAll > 0 ? nfrAll - 1 : 0;
        const uint32_t pLo = seg_begin(seg), pEnd = seg_begin(seg + 1);
        segFrame0 = pLo > A.seg_warm ? pLo - A.seg_warm : 0u;                      // (uniform)
        if (seg > 0 && pLo >= nper) nfr = 0;                                       // the voice ended before this segment
        else if (nfrAll > 0) {
            const uint32_t pHi = pEnd < nper ? pEnd : nper;
            nfr = pHi - segFrame0 + 1;
            segLast = pHi == nper;
            segOutEnd = outputs_before((uint64_t)pHi * CP);                        // (used when the voice goes on)
        }
    }
    const uint32_t nfrMax = wave_max_u32(nfr);          // same 64 voices in every wave of the group
    const uint32_t ntubeMax = nfrMax > 0 ? (nfrMax - 1) * CP : 0;
    // (nfrMax-1) control periods, then the converter's 2*pad zero flush (TRMRingBuffer.m:85-93).
    // Lanes whose utterance is shorter than the group's longest keep stepping on their last frame;
    // the tube stage hands zeros to the converter past a voice's own end.
    const bool sFirst = !kStream || (A.stream_flags & 1u), sLast = !kStream || (A.stream_flags & 2u);
    const bool sHold = kStream && (A.stream_flags & 4u);       // TRAcT's loop order: a period runs on the frame that ends it, held
    // (segments: a workgroup's lanes either end inside the segment -- their flush follows -- or run to its end: nTotal
    // carries the flush's 2*pad samples either way, lanes that go on stop emitting at segOutEnd)
    const uint32_t nBase = kStream ? A.stream_n_base : kSeg ? segFrame0 * CP : 0u;
    const uint32_t kBase = kStream ? A.stream_k_base : kSeg ? outputs_before((uint64_t)seg_begin(seg) * CP) : 0u;
    // this voice's state record: per workgroup a block of [kStreamFloats fields][64 lanes] floats -- a wave's 64 lanes touch
    // 64 consecutive floats per field (voice-major records cost 64 cache lines per field and instruction) and a field is a
    // CONSTANT 256 bytes from the one before (one base address per lane: per-field 64-bit strides cost the streaming
    // instance its registers -- it spilled)
    struct StateRef {
        float *base;
        // (the opaque copy keeps the address arithmetic inside the rare block that uses it: hoisted out of the step loop
        // the 28 store addresses of a save were spilled to scratch)
        __device__ float &operator[](int i) const
        {
            float *b = base;
            asm volatile("" : "+v"(b));
            return b[i * kWave];
        }
    };
    const StateRef st{kStream ? A.stream_state + (size_t)wg * kStreamFloats * kWave + lane : nullptr};
    auto st_load_f64 = [&]() {
        const unsigned long long lo = __builtin_bit_cast(unsigned, st[0]), hi = __builtin_bit_cast(unsigned, st[1]);
        return __builtin_bit_cast(double, (hi << 32) | lo);
    };
    auto st_store_f64 = [&](double x) {
        const unsigned long long b = __builtin_bit_cast(unsigned long long, x);
        st[0] = __builtin_bit_cast(float, (unsigned)b);
        st[1] = __builtin_bit_cast(float, (unsigned)(b >> 32));
    };
    const uint32_t nTotal = nfrMax > 0 ? ntubeMax + (sLast ? 2u * (uint32_t)C.padSize : 0u) : 0;
    // Output rows (64 lanes) a step's kTB tube samples turn into: kTB * 2^16/inc; the convert waves get
    // strictly more than that per step, in row pairs, split between the two waves.
    // the last tube block is written at step ceil(nTotal/kTB), readable one step later; the convert waves
    // are never more than one block (16 pairs, 8 per wave) behind
    // (8 row pairs per wave per block, metered at just over the production rate: a block takes about as
    // long as the 32*inc/2^16 tube samples it spans)
    const uint32_t nSteps = nTotal > 0 ? (nTotal + kTB - 1) / kTB + 3 + 2 * ((kCvtCols * inc / 65536u) / kTB + 2) + 4 : 0;
    // a voice without frames (a silent no-op, TRMTubeModel.m:274-277) reads row 0 of the buffer
    const float *frames = A.frames + (nfr > 0 ? (A.frame_offset[v] + segFrame0) * 16 : 0);
    const uint32_t ntubeLane = nfr > 0 ? (nfr - 1) * CP : 0;

    for (int i = threadIdx.x; i < kWave * kYStride; i += kWave * kRoles) sY[i] = 0.0f;   // 25 zeros of pre-roll
    if (kStream && !sFirst) {
        // the last 32 tube samples of the chunks before this one, at their places in the ring (global sample G at slot G + 25)
        __syncthreads();
        for (int i = threadIdx.x; i < kWave * 32; i += kWave * kRoles) {
            const int q = i & (kWave - 1), t = i >> 6;
            const float y = A.stream_state[((size_t)wg * kStreamFloats + 72 + t) * kWave + q];
            const uint32_t slot = (nBase - 32u + (uint32_t)t + (kSrcWindow - 1)) & (kYRing - 1);
            sY[q * kYStride + slot] = y;
            if (slot < (uint32_t)kYMirror) sY[q * kYStride + slot + kYRing] = y;
        }
    }
    __syncthreads();

    if (role == 0) {
        // ------------------------------------------------------------ osc: tracks + oscillator, block i at step i
        __builtin_amdgcn_s_setprio(kStream ? TRM_PRIO_OSC_S : TRM_PRIO_OSC);
        auto sine = [&](int i) { return sine_table(i); };
        OscState S;
        ExciteTrack T;
        S.oscPos = (kStream && !sFirst) ? st_load_f64() : 0.0;
        if (kSeg) {
            // the oscillator's position at the warm-up start: the wrapped advances between the warm-up starts of the
            // segments so far, summed in order (every term and sum a multiple of 2^-30 below 2^10: exact)
            const double *ph = A.seg_phase + vRaw;
            const size_t pitch = (size_t)A.seg_wg_per_seg * kWave;
            for (uint32_t q = 1; q <= seg; q++) {
                const double t = S.oscPos + ph[q * pitch];
                S.oscPos = t > 511.0 ? t - 512.0 : t;
            }
        }
        // (both frames of a control period are fetched when it starts, once per ~80 samples: carrying the current frame
        // to the next boundary in registers costs a register-to-register copy of it per STEP, the loop's phi nodes)
        auto frame_at = [&](uint32_t i) { return nfr > 0 ? (i < nfr ? i : nfr - 1) : 0u; };
        uint32_t j = CP, f = 0;
        STAMP_DECL
        for (uint32_t step = 0; step < nSteps; step++) {
            STAMP_BEGIN
            const int buf = step & 1;
            for (int u = 0; u < kTB; u++) {
                const uint32_t n = step * kTB + u;
                if (n < nTotal) {
                    if (j == CP) {   // -setControlRateParameters:previous: (TRMTubeModel.m:289)
                        j = 0;
                        f++;
                        float prev[4], cur[4];
                        load_frame(frames, frame_at(sHold ? f : f - 1), prev, 1);
                        load_frame(frames, frame_at(f), cur, 1);
                        excite_track_setup(T, C, prev, cur);
                    }
                    OscOut O = osc_sample(S, T, C, (int)j, sine);
                    if (kStream && n + 1u == ntubeLane && laneValid) st_store_f64(S.oscPos);   // the chunk's last sample
                    j++;
                    sW[(buf * kTB + u) * kWave + lane] = make_float4(O.wa, O.wb, O.ax, O.ah1);
                }
            }
            STAMP_MID
            step_barrier();
            STAMP_END
        }
        STAMP_STORE(role)
    } else if (role == 1) {
        // ------------------------------------------------------------ mix: FIR + noise mixing, block i-1 at step i
        __builtin_amdgcn_s_setprio(TRM_PRIO_MIX);
        const float *const lpNoise = A.lp_noise + (kSeg ? nBase : 0u);      // (a stream's pointer arrives advanced)
        auto fill_noise_half = [&](uint32_t nFirst, int half) {
            dma4(lpNoise + nFirst + lane, &sNoise[half * kNoiseHalf]);
        };
        FirState S;
        for (int i = 0; i < 24; i++) S.fir[i] = (kStream && !sFirst) ? st[2 + i] : 0.f;
        // the throat's one-pole low-pass runs here (tube_step<., kThroatDone>): it sees the excitation only
        float throatY = (kStream && !sFirst) ? st[64] : 0.f;
        // The 25 distinct FIR taps live in VGPRs of this wave (uniform values): as SGPRs they would
        // exceed the scalar file together with the other constants and be spilled to VGPR lanes.
        float firv[kFirUnique];
        for (int i = 0; i < kFirUnique; i++) asm volatile("v_mov_b32 %0, %1" : "=v"(firv[i]) : "s"(C.fir[i]));
        if (nSteps > 0) {
            fill_noise_half(0, 0);
            fill_noise_half(kNoiseHalf, 1);
            dma_wait_all();
        }
        STAMP_DECL
        for (uint32_t step = 0; step < nSteps; step++) {
            STAMP_BEGIN
            if (step >= 1) {
                const uint32_t blk = step - 1;
                const int buf = blk & 1;
                for (int u = 0; u < kTB; u++) {
                    const uint32_t n = blk * kTB + u;
                    if (n < nTotal) {
                        if ((n & (kNoiseHalf - 1)) == 0 && n > 0) {
                            // entering a noise half: it was requested one half ago; refill the other half
                            dma_wait_all();
                            fill_noise_half(n + kNoiseHalf, ((n / kNoiseHalf) + 1) & 1);
                        }
                        const float4 w = sW[(buf * kTB + u) * kWave + lane];
                        OscOut O;
                        O.wa = w.x; O.wb = w.y; O.ax = w.z; O.ah1 = w.w;
                        Excitation E = mix_sample(S, C, firv, O, sNoise[n & (kNoiseRing - 1)]);
                        throatY = throat_filter(throatY, C.ta0, C.tb1, E.thr);
                        if (kStream && n + 1u == ntubeLane && laneValid) {
                            for (int i = 0; i < 24; i++) st[2 + i] = S.fir[i];
                            st[64] = throatY;
                        }
                        sX[(buf * kTB + u) * kWave + lane] = make_float4(E.gin, E.sig, throatY, 0.0f);
                    }
                }
            }
            STAMP_MID
            step_barrier();
            STAMP_END
        }
        STAMP_STORE(role)
        dma_wait_all();   // nothing may still be writing LDS when the wave ends
    } else if (role == 2 || role == 3) {
        // ------------------------------------------------------------ coef (this wave: samples of parity u), block i-1 at step i
        const int u = role - 2;
        __builtin_amdgcn_s_setprio(TRM_PRIO_COEF);
        CoefTrack T;
        // the stage's constants as vector registers (cf. the tube wave's TubeConst): ~14 of the stage's 119 instructions per
        // sample read one, and with a scalar operand they issue at the slow rate (VERDICT r03 asked for this A/B in the
        // throughput waves: 65 536 voices 16.03 -> 15.87 ms, -1.0 %, profiles/ab_r04.txt)
        CoefConst CC;
        {
            auto vcopy = [](float s) { float r; asm volatile("v_mov_b32 %0, %1" : "=v"(r) : "s"(s)); return r; };
            CC.damping = vcopy(C.damping); CC.apScaleSq = vcopy(C.apScaleSq); CC.mA10 = vcopy(C.mA10);
            CC.noseR1sq = vcopy(C.noseR1sq); CC.invSampleRate = vcopy(C.invSampleRate); CC.fricGain = vcopy(C.fricGain);
        }
        auto frame_at = [&](uint32_t i) { return nfr > 0 ? (i < nfr ? i : nfr - 1) : 0u; };
        // position of this wave's next sample in its control period; the first sample starts period 1
        uint32_t j = CP + (uint32_t)u, f = 0;
        STAMP_DECL
        for (uint32_t step = 0; step < nSteps; step++) {
            STAMP_BEGIN
            const int buf = (step + 1) & 1;             // == (step - 1) & 1
            const uint32_t n = (step - 1) * kTB + u;    // step 0: wraps past nTotal, no work
            if (step >= 1 && n < nTotal) {
                if (j >= CP) {      // the host guarantees CP >= 2*kTB: at most one period boundary per step
                    j -= CP;
                    f++;
                    float prev[16], cur[16];       // (fetched per period, not carried: see the oscillator wave)
                    load_frame(frames, frame_at(sHold ? f : f - 1), prev, 4);
                    load_frame(frames, frame_at(f), cur, 4);
                    coef_track_setup(T, C, prev, cur);
                }
                Coefs K = coef_sample<kStream>(T, CC, (int)j);
                j += kTB;
                // 20 floats per sample: C8, alphaLR and bpAlpha are re-derived by the tube wave (1 op each)
                float4 *dst = &sK[((buf * kTB + u) * kKQuads) * kWave + lane];
                // (the junction coefficients travel as (1 + k) * damping: tube_step's working form)
                dst[0 * kWave] = make_float4(K.td[0], K.td[1], K.td[2], K.td[3]);
                dst[1 * kWave] = make_float4(K.td[4], K.td[5], K.td[6], K.onePlusK8);
                dst[2 * kWave] = make_float4(K.alphaU, K.ntd1, K.bpBeta, K.bpGamma);
                dst[3 * kWave] = make_float4(K.tap[0], K.tap[1], K.tap[2], K.tap[3]);
                dst[4 * kWave] = make_float4(K.tap[4], K.tap[5], K.tap[6], K.tap[7]);
            }
            STAMP_MID
            step_barrier();
            STAMP_END
        }
        STAMP_STORE(role)
    } else if (role == 4) {
        // ------------------------------------------------------------ tube: block i-2 at step i
        // two wave sets: sample 2i steps wA -> wB, sample 2i+1 steps wB -> wA (kTB == 2: no state copies)
        __builtin_amdgcn_s_setprio(kStream ? TRM_PRIO_TUBE_S : TRM_PRIO_TUBE);
        Waves wA, wB;
        TubeFilters F;
        waves_reset(wA);
        waves_reset(wB);
        filters_reset(F);
        if (kStream && !sFirst) {
            for (int i = 0; i < 10; i++) { wA.oT[i] = st[26 + i]; wA.oB[i] = st[36 + i]; }
            for (int i = 0; i < 6; i++) { wA.nT[i] = st[46 + i]; wA.nB[i] = st[52 + i]; }
            F.mReflY = st[58]; F.mRadX = st[59]; F.mRadY = st[60]; F.nReflY = st[61]; F.nRadX = st[62]; F.nRadY = st[63];
            F.bpX1 = st[65]; F.bpX2 = st[66]; F.bpY1 = st[67]; F.bpY2 = st[68];       // (st[64], the throat's memory: the mix wave)
        }
        // the step's constants as vector registers (tube_step): opaque moves, so that the compiler cannot go back to the
        // kernel arguments' scalar registers.  (With the other round-3 changes 18.2 ms against 18.5 ms for the saturating
        // batch, profiles/ab_r03.txt; on the round-2 kernel alone the same change measured 1 % slower.)
        TubeConst TC;
        float mA10v, negMA10v;
        {
            auto vcopy = [](float s) { float r; asm volatile("v_mov_b32 %0, %1" : "=v"(r) : "s"(s)); return r; };
            TC.damping = vcopy(C.damping); TC.mCoeff = vcopy(C.mCoeff); TC.nCoeff = vcopy(C.nCoeff);
            for (int i = 0; i < 4; i++) TC.nasalTd[i] = vcopy(C.nasalTd[i]);
            TC.nasalK6a = vcopy(C.nasalK6a); TC.onePlusNK6 = vcopy(C.onePlusNK6);
            TC.ta0 = TC.tb1 = 0.0f;        // (the throat section runs in the mix wave)
            TC.throatGain = vcopy(C.throatGain);
            mA10v = vcopy(C.mA10);
            negMA10v = vcopy(-C.mA10);
        }
        float *const ring = &sY[lane * kYStride];
        // down-sampling batches: tube-rate samples (and the zero flush) go to HBM for trm_downsample_kernel
        float *const tubeOut = A.tube_out ? A.tube_out + A.tube_offset[v] : nullptr;
        auto one = [&](const Waves &o, Waves &nw, int buf, int u, uint32_t n) {
            const float4 x = sX[(buf * kTB + u) * kWave + lane];
            const float4 *src = &sK[((buf * kTB + u) * kKQuads) * kWave + lane];
            const float4 k0 = src[0 * kWave], k1 = src[1 * kWave], k2 = src[2 * kWave];
            const float4 t0 = src[3 * kWave], t1 = src[4 * kWave];
            Excitation E;
            E.gin = x.x; E.sig = x.y; E.thr = x.z;
            Coefs K;
            K.td[0] = k0.x; K.td[1] = k0.y; K.td[2] = k0.z; K.td[3] = k0.w;
            K.td[4] = k1.x; K.td[5] = k1.y; K.td[6] = k1.z; K.onePlusK8 = k1.w;
            K.k8a = fma_f(K.onePlusK8, mA10v, negMA10v); // C8 a10 = (1 + C8) a10 - a10 in one operation (C8 is near -1 when the mouth closes: no cancellation here)
            K.alphaU = k2.x; K.ntd1 = k2.y; K.bpBeta = k2.z; K.bpGamma = k2.w;
            K.alphaLR = fma_f(-0.5f, K.alphaU, 1.0f);    // the three alphas sum to 2 (TRMTubeModel.m:733-736)
            K.bpAlpha = fma_f(-0.5f, K.bpBeta, 0.25f);   // (1/2 - beta) / 2, TRMFilters.m:16 (the halving is exact: the same bits in one operation)
            K.tap[0] = t0.x; K.tap[1] = t0.y; K.tap[2] = t0.z; K.tap[3] = t0.w;
            K.tap[4] = t1.x; K.tap[5] = t1.y; K.tap[6] = t1.z; K.tap[7] = t1.w;
            K.pad_ = 0.0f;
            float y = tube_step<TubeConst, true>(o, nw, F, TC, E, K);
            if (kStream && n + 1u == ntubeLane && laneValid) {       // the chunk's last sample: what the next chunk starts from
                for (int i = 0; i < 10; i++) { st[26 + i] = nw.oT[i]; st[36 + i] = nw.oB[i]; }
                for (int i = 0; i < 6; i++) { st[46 + i] = nw.nT[i]; st[52 + i] = nw.nB[i]; }
                st[58] = F.mReflY; st[59] = F.mRadX; st[60] = F.mRadY; st[61] = F.nReflY; st[62] = F.nRadX; st[63] = F.nRadY;
                st[65] = F.bpX1; st[66] = F.bpX2; st[67] = F.bpY1; st[68] = F.bpY2;
            }
            y = n < ntubeLane ? y : 0.0f;      // zero flush / voices shorter than the group's longest
            // converter position of tube sample n is n + 25 (25 zeros of pre-roll); a chunk's sample n is the
            // utterance's sample nBase + n
            const uint32_t slot = (nBase + n + (kSrcWindow - 1)) & (kYRing - 1);
            ring[slot] = y;
            if (slot < (uint32_t)kYMirror) ring[slot + kYRing] = y;   // mirror: windows never wrap
            if (kSeg) {
                // a segment writes its own stretch of the voice's tube-rate row (global index nBase + n), not its warm-up's;
                // the voice's last segment appends the flush
                const uint32_t gn = nBase + n;
                if (tubeOut && laneValid && gn >= seg_begin(seg) * CP && n < ntubeLane + (segLast ? 2u * (uint32_t)C.padSize : 0u))
                    tubeOut[gn] = y;
            } else if (tubeOut && laneValid && n < ntubeLane + (sLast ? 2u * (uint32_t)C.padSize : 0u)) tubeOut[n] = y;
        };
        static_assert(kTB == 2, "the tube stage ping-pongs two wave sets per step");
        STAMP_DECL
        for (uint32_t step = 0; step < nSteps; step++) {
            STAMP_BEGIN
            if (step >= 2 && (step - 2) * kTB < nTotal) {
                // both samples of the block are stepped even when the second one lies past nTotal (odd
                // totals): its inputs are stale LDS contents, its output is forced to 0 and never read
                const uint32_t blk = step - 2;
                const int buf = blk & 1;
                one(wA, wB, buf, 0, blk * kTB);
                one(wB, wA, buf, 1, blk * kTB + 1);
            }
            STAMP_MID
            step_barrier();
            STAMP_END
        }
        STAMP_STORE(role)
        if (kStream && laneValid && ntubeLane > 0) {
            // the last 32 tube samples so far (older ones than the chunk's are still in the ring, where the start put them)
            for (int t = 0; t < 32; t++) st[72 + t] = ring[(nBase + ntubeLane - 32u + (uint32_t)t + (kSrcWindow - 1)) & (kYRing - 1)];
        }
    } else {
        // ------------------------------------------------------------ convert (lane = output time)
        const int cw = role - 5;                    // this wave converts voices 32*cw .. 32*cw+31
        __builtin_amdgcn_s_setprio(kStream ? TRM_PRIO_CVT_S : TRM_PRIO_CVT);
        // outputs of this launch per voice: the utterance's (TRMSampleRateConverter.m:160-173); a chunk's: global indices
        // kBase <= k < stream_k_end, the same for every voice
        uint32_t noutLane = 0;
        if (nfr > 0) {
            uint64_t total = (uint64_t)ntubeLane + 2ull * (uint32_t)C.padSize;
            noutLane = (uint32_t)((total * 65536ull + inc - 1) / inc);
        }
        if (kStream) noutLane = A.stream_k_end - kBase;
        uint32_t noutAll = 0;                   // (segments: the whole utterance's count)
        if (kSeg) {
            if (nfrAll > 0) noutAll = (uint32_t)((((uint64_t)(nfrAll - 1) * CP + 2ull * (uint32_t)C.padSize) * 65536ull + inc - 1) / inc);
            noutLane = nfr > 0 ? (segLast ? noutAll : segOutEnd) - kBase : 0u;
        }
        if (!laneValid) noutLane = 0;
        // per-voice values stay in the VGPRs of lane == voice and are broadcast per row with v_readlane
        const uintptr_t myOut = reinterpret_cast<uintptr_t>(A.out + A.out_offset[v] + (kSeg ? kBase : 0u));
        const uint32_t myLo = (uint32_t)myOut, myHi = (uint32_t)(myOut >> 32);
        const uint32_t noutMax = wave_max_u32(noutLane);
        // (a down-sampling batch is converted by trm_downsample_kernel: no blocks here, only the barriers)
        const uint32_t nBlocks = C.upsample ? (noutMax + kCvtCols - 1) / kCvtCols : 0;
        const int col = lane & (kCvtCols - 1);      // output within the block
        const bool upper = lane >= kCvtCols;        // which of a row's two voices
        // destination pointer and length of every voice of this wave, laid out per (row, half) so that a row's
        // lanes fetch theirs with one broadcast ds_read_b128: {length, ptr lo, ptr hi, -}
        uint4 *const info = &sInfo[cw * 32];
        if (lane >= 32 * cw && lane < 32 * cw + 32) info[lane - 32 * cw] = make_uint4(noutLane, myLo, myHi, 0u);
        // running max |y| (TRMSampleRateConverter.m:206-208) per (row of this wave, lane) in LDS: row r covers
        // voice 32*cw + 2*r (+1 in the upper half); folded across the 32 columns once, at the end
        float *const mxTile = &sMx[cw * 16 * kWave];
        for (int r = 0; r < 16; r++) mxTile[r * kWave + lane] = 0.0f;

        // This lane's output reads a 16-byte ALIGNED 32-sample window (a lone wave issues wide LDS reads at
        // full rate, narrow ones at a fraction of it): the window starts winOff = e & 3 samples early and
        // the lane's 26 coefficients are fetched shifted right by winOff.  Coefficient rows are 32 floats
        // (26 + 6 zeros) with 4 zeros in front of row 0, so the shifted fetch only ever picks up zeros.
        v2f cc[16];
        auto fetch_row = [&](uint32_t blk) {
            const uint32_t k = kBase + blk * kCvtCols + col;
            const uint32_t off = src_position(k, inc) & 3u;
            const float *pc = A.src_rows + (size_t)src_phase(k, inc) * kSrcRowC - off;
            // (the loads land while the wave idles until its next metered pair)
            for (int q = 0; q < 15; q++) cc[q] = v2f{pc[2 * q], pc[2 * q + 1]};      // (cc[15]: always zeros, unused)
        };
        uint32_t blk = 0, pr = 0;       // next work item: row pair `pr` (0..7) of block `blk`: voices 32*cw + 4*pr .. +3
        uint32_t winBase = 0;           // this lane's aligned window start inside a voice's ring (floats)
        uint32_t kLane = 0;             // this lane's output index (within the launch; + kBase: within the utterance)
        uint32_t needLast = 0;          // last tube sample the current block reads (uniform)
        uint32_t needNext = 0;          // ... and the one after it: complete already = this wave is a block behind
        auto begin_block = [&]() {
            kLane = blk * kCvtCols + col;
            winBase = src_position(kBase + kLane, inc) & (kYRing - 1) & ~3u;
            // highest tube sample (of this launch) the block reads: the window of output k ends at tube sample e_k; outputs
            // past the longest voice's end are masked, so the last block only waits for the final sample
            needLast = src_position(kBase + blk * kCvtCols + (kCvtCols - 1), inc) - nBase;
            needLast = needLast < nTotal - 1 ? needLast : nTotal - 1;
            needNext = src_position(kBase + (blk + 1) * kCvtCols + (kCvtCols - 1), inc) - nBase;     // (past the end: never "behind")
            fetch_row(blk);
        };
        if (nBlocks > 0) begin_block();
        typedef __attribute__((address_space(1))) float *GlobalFloatPtr;
        typedef __attribute__((address_space(3))) float *LdsFloatPtr;
        // Work is metered so that it spreads evenly over the steps: each step earns `earn` (16.16) row pairs,
        // a little more than the kTB tube samples of a step turn into (pairs of this wave per step =
        // kTB * 2^16/inc / 4), and a pair runs when a whole one has been earned and its block is readable.
        const uint32_t earn = (uint32_t)(((uint64_t)kTB << 32) / inc / 4) + 2048;
        // the cap must leave room to catch up after waiting for a block (a cap of about `earn` loses credit while
        // it waits and the wave falls behind until the ring laps it: seen at a 30 cm tube, ratio 3.8)
        const uint32_t capPairs = (earn + 0x18000u) >> 16;                  // floor(earn + 1.5)
        const uint32_t creditCap = (capPairs > 2u ? capPairs : 2u) << 16;   // two pairs per step at speech rates (earn ~ 1.1)
        uint32_t credit = 0;
        SUB_DECL
        auto do_pair = [&]() {
            SUB_START
            // two wave-rows at a time = four voices: their LDS reads and FMA chains overlap
            const int la = 4 * (int)pr, lb = la + 2;           // voices local to this wave
            const int va = 32 * cw + la, vb = 32 * cw + lb;
            const int ha = upper ? 1 : 0;
            const float4 *wa = reinterpret_cast<const float4 *>(&sY[(va + ha) * kYStride + winBase]);
            const float4 *wb = reinterpret_cast<const float4 *>(&sY[(vb + ha) * kYStride + winBase]);
            float4 qa[8], qb[8];
            for (int q = 0; q < 8; q++) { qa[q] = wa[q]; qb[q] = wb[q]; }
            const uint4 ia = info[la + ha], ib = info[lb + ha];
            SUB_LAP(0)
            // 32-term dot products as packed FMAs: (even, odd) partial sums, two chains per row
            v2f a0 = v2f{qa[0].x, qa[0].y} * cc[0], a1 = v2f{qa[0].z, qa[0].w} * cc[1];
            v2f b0 = v2f{qb[0].x, qb[0].y} * cc[0], b1 = v2f{qb[0].z, qb[0].w} * cc[1];
            for (int q = 1; q < 7; q++) {
                a0 = __builtin_elementwise_fma(v2f{qa[q].x, qa[q].y}, cc[2 * q], a0);
                a1 = __builtin_elementwise_fma(v2f{qa[q].z, qa[q].w}, cc[2 * q + 1], a1);
                b0 = __builtin_elementwise_fma(v2f{qb[q].x, qb[q].y}, cc[2 * q], b0);
                b1 = __builtin_elementwise_fma(v2f{qb[q].z, qb[q].w}, cc[2 * q + 1], b1);
            }
            // (terms 30, 31 are always zeros: the row is 26 coefficients shifted by at most 3)
            a0 = __builtin_elementwise_fma(v2f{qa[7].x, qa[7].y}, cc[14], a0);
            b0 = __builtin_elementwise_fma(v2f{qb[7].x, qb[7].y}, cc[14], b0);
            a0 += a1;
            b0 += b1;
            const float ya = a0.x + a0.y, yb = b0.x + b0.y;
            SUB_LAP(2)
            const bool okA = kLane < ia.x, okB = kLane < ib.x;
            if (okA) reinterpret_cast<GlobalFloatPtr>(((uintptr_t)ia.z << 32) | ia.y)[kLane] = ya;
            if (okB) reinterpret_cast<GlobalFloatPtr>(((uintptr_t)ib.z << 32) | ib.y)[kLane] = yb;
            SUB_LAP(3)
            // running max |y| per (row, lane): conflict-free LDS float-max, folded across columns at the end
            __builtin_amdgcn_ds_fmaxf((LdsFloatPtr)&mxTile[(2 * pr) * kWave + lane], okA ? fabsf(ya) : 0.0f, 0, 0, false);
            __builtin_amdgcn_ds_fmaxf((LdsFloatPtr)&mxTile[(2 * pr + 1) * kWave + lane], okB ? fabsf(yb) : 0.0f, 0, 0, false);
            if (++pr == 8) {
                pr = 0;
                blk++;
                if (blk < nBlocks) begin_block();
            }
            SUB_LAP(4)
        };
        STAMP_DECL
        for (uint32_t step = 0; step < nSteps; step++) {
            STAMP_BEGIN
            // visible after the previous barrier: tube samples n < (step-2)*kTB
            const uint32_t ready = step >= 2 ? (step - 2) * kTB : 0;
            credit += earn;
            if (credit > creditCap) credit = creditCap;     // a ready block is spread over the next steps, not done in a burst
            // Metered by the credit (smooth: at most creditCap pairs in a step) -- but a wave that is a whole block
            // behind production works through its backlog regardless: the credit alone loses what it earns while it
            // waits for a block to complete, and at some rate ratios that slowly let the ring lap the converter.
            while (blk < nBlocks && needLast < ready && (credit >= (1u << 16) || needNext < ready)) {
                credit = credit >= (1u << 16) ? credit - (1u << 16) : credit;
                do_pair();
            }
            STAMP_MID
            step_barrier();
            STAMP_END
        }
        STAMP_STORE(role)
        SUB_STORE(role)
        // every tube sample is in the ring now: anything still queued needs no further hand-off
        while (blk < nBlocks) do_pair();
        // fold the running maxima across the 32 columns of each half: row r -> voices 32*cw + 2r (lanes
        // 0-31) and 32*cw + 2r + 1 (lanes 32-63)
        float myMax = 0.0f;     // collected by lanes 0..31: voice 32*cw + lane
#pragma unroll
        for (int r = 0; r < 16; r++) {
            float m = mxTile[r * kWave + lane];
            for (int off = 16; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, kWave));
            const float lowHalf = __shfl(m, 0, kWave), highHalf = __shfl(m, 32, kWave);
            if (lane == 2 * r) myMax = lowHalf;
            if (lane == 2 * r + 1) myMax = highHalf;
        }
        const uint32_t ov = vblock * kWave + 32 * cw + (lane & 31);
        const uint32_t nov = __builtin_amdgcn_ds_bpermute(4 * (32 * cw + (lane & 31)), kSeg ? noutAll : noutLane);
        if (lane < 32 && ov < A.nvoices && C.upsample) {
            if (kSeg) {
                // (non-negative floats order like their bit patterns; max_sample was zeroed by the launcher)
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

// Prefix pass of a time-split launch (TubeArgs::seg_*), two kernels.
//   trm_phase_period_kernel   thread (v, p): the oscillator's advance over voice v's control period p, in units of 2^-30 table
//       entries -- the same track set-up and increments as osc_sample (TRMWavetable.m:165-181, osc_increment), summed as
//       integer-valued doubles (every term < 2^35, 79-200 of them: exact) and reduced modulo the table (512 * 2^30) --
//       to period_adv[v * max_nframes + p]; and the guard: a frame whose frication bandwidth lies below bw_floor sets *gate
//       (that band-pass remembers longer than the warm-up: the batch runs as whole utterances instead, TubeArgs::gate).
//   trm_phase_segment_kernel  thread (q, v): the sum of period_adv over the periods between the warm-up starts of segments
//       q and q + 1, as a table position in (-1, 511] (mod0's range, TRMWavetable.m:28-34), to seg_phase[q + 1][v]; the tube
//       kernel's segment s sums entries 1 .. s.  Sums and wraps of exact values: the position a segment starts from is
//       bit for bit the one the uninterrupted oscillator has there.
// (Round 4's first version walked every segment's periods serially in one thread: 0.33-0.43 ms of a 2.7-3.3 ms launch.)
constexpr double kPhaseModulus = 549755813888.0;       // 512 * 2^30
__global__ __launch_bounds__(256) void trm_phase_period_kernel(const Const C, const PhaseArgs P)
{
    const uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t pitch = P.max_nframes;              // periods per voice in period_adv (max_nframes - 1 used)
    const uint32_t v = idx / pitch, p = idx - v * pitch;
    if (v >= P.nvoices) return;
    const uint32_t nfr = min(P.nframes[v], P.max_nframes);
    if (p + 1 >= nfr) {
        // (no period here; the voice's last frame still counts for the guard)
        if (nfr > 0 && p + 1 == nfr && (P.frames + P.frame_offset[v] * 16)[(size_t)p * 16 + 6] < P.bw_floor) atomicOr(P.gate, 1u);
        return;
    }
    const float *frames = P.frames + P.frame_offset[v] * 16;
    float prev[4], cur[4];
    load_frame(frames, p, prev, 1);
    load_frame(frames, p + 1, cur, 1);
    if (frames[(size_t)p * 16 + 6] < P.bw_floor) atomicOr(P.gate, 1u);
    ExciteTrack T;
    excite_track_setup(T, C, prev, cur);
    double sum = 0.0;
    const uint32_t CP = (uint32_t)C.controlPeriod;
    for (uint32_t j = 0; j < CP; j++) {
        // osc_increment's rounded product, left in its integer units: (f0 * 0.5) * basicIncrement * 2^30 -- the factors
        // 0.5 and 2^30 are exact, so this is the same rounding of f0 * basicIncrement
        sum += rint_d(((T.f0 * 0.5) * C.basicIncrement) * 1073741824.0);
        T.f0 *= T.f0Ratio;
    }
    sum += sum;                                         // two increments per tube sample (2x oversampled oscillator, :178-181)
    sum -= kPhaseModulus * floor(sum * (1.0 / kPhaseModulus));
    sum = sum < 0.0 ? sum + kPhaseModulus : sum;        // (the quotient's rounding at an exact multiple)
    sum = sum >= kPhaseModulus ? sum - kPhaseModulus : sum;
    P.period_adv[(size_t)v * pitch + p] = sum;
}

__global__ __launch_bounds__(256) void trm_phase_segment_kernel(const Const C, const PhaseArgs P)
{
    const uint32_t lanes = P.seg_wg_per_seg * P.voices_per_wg;      // (the tube kernel's index pitch: 64 voices per workgroup, or 16)
    const uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t q = idx / lanes, v = idx - q * lanes;
    if (q + 1 >= P.nseg || v >= P.nvoices) return;
    const uint32_t nfr = min(P.nframes[v], P.max_nframes);
    const uint32_t nper = nfr > 0 ? nfr - 1 : 0;
    auto warm_start = [&](uint32_t sgm) {
        const uint32_t p = sgm == 0 ? 0u : P.seg_first + (sgm - 1) * P.seg_periods;
        return p > P.seg_warm ? p - P.seg_warm : 0u;
    };
    uint32_t lo = warm_start(q), hi = warm_start(q + 1);
    lo = lo < nper ? lo : nper;
    hi = hi < nper ? hi : nper;
    const double *adv = P.period_adv + (size_t)v * P.max_nframes;
    double sum = 0.0;
    for (uint32_t p = lo; p < hi; p++) {
        sum += adv[p];                                  // (integers below 2^39, at most a few thousand of them: exact)
        sum = sum >= kPhaseModulus ? sum - kPhaseModulus : sum;
    }
    double pos = sum * (1.0 / 1073741824.0);            // exact: [0, 512)
    pos = pos > 511.0 ? pos - 512.0 : pos;              // mod0's representative
    P.seg_phase[(size_t)(q + 1) * lanes + v] = pos;
}

// The launch order of a time-split grid.  A ragged batch leaves most (segment, block of voices) pairs of the rectangular grid
// without work -- the block's voices ended before the segment -- and a workgroup that exits at once does NOT hand its place
// to the next one in line: on this hardware the 513th workgroup of a grid of 528 started when the first long-running ones
// ended, 2.5 ms late, although 224 of the first 512 had exited within a microsecond (tools/split_birth_probe.py).  So the
// pairs with work go first: trm_seg_blocks_kernel finds every block's longest voice, trm_seg_map_kernel (one workgroup) lists
// the pairs with work in (segment, block) order and the others after them.
__global__ __launch_bounds__(256) void trm_seg_blocks_kernel(const PhaseArgs P)
{
    // thread = voice; a block's 64 (or 16) voices are neighbouring lanes of one wave
    const uint32_t v = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t longest = v < P.nvoices ? min(P.nframes[v], P.max_nframes) : 0u;
    for (uint32_t off = P.voices_per_wg >> 1; off > 0; off >>= 1) longest = max(longest, (uint32_t)__shfl_xor((int)longest, (int)off, kWave));
    const uint32_t blk = v / P.voices_per_wg;
    if ((v & (P.voices_per_wg - 1)) == 0 && blk < P.seg_wg_per_seg) P.block_frames[blk] = longest;
}

constexpr int kMapThreads = 1024;
__global__ __launch_bounds__(kMapThreads) void trm_seg_map_kernel(const PhaseArgs P)
{
    __shared__ uint32_t sCount[kMapThreads];
    const uint32_t n = P.nseg * P.seg_wg_per_seg;
    const uint32_t chunk = (n + kMapThreads - 1) / kMapThreads;
    const uint32_t lo = min(threadIdx.x * chunk, n), hi = min(lo + chunk, n);
    // (what the tube kernels decide per lane: segment 0 always runs; a later one when a voice of the block reaches it)
    auto has_work = [&](uint32_t i) {
        const uint32_t sgm = i / P.seg_wg_per_seg, blk = i - sgm * P.seg_wg_per_seg;
        if (sgm == 0) return true;
        const uint32_t nfr = P.block_frames[blk], nper = nfr > 0 ? nfr - 1 : 0;
        return P.seg_first + (sgm - 1) * P.seg_periods < nper;
    };
    uint32_t mine = 0;
    for (uint32_t i = lo; i < hi; i++) mine += has_work(i) ? 1u : 0u;
    sCount[threadIdx.x] = mine;
    __syncthreads();
    // inclusive scan over the threads' counts
    for (uint32_t off = 1; off < (uint32_t)kMapThreads; off <<= 1) {
        const uint32_t add = threadIdx.x >= off ? sCount[threadIdx.x - off] : 0u;
        __syncthreads();
        sCount[threadIdx.x] += add;
        __syncthreads();
    }
    const uint32_t total = sCount[kMapThreads - 1];
    uint32_t busy = sCount[threadIdx.x] - mine;             // pairs with work before this thread's chunk
    for (uint32_t i = lo; i < hi; i++) {
        const uint32_t sgm = i / P.seg_wg_per_seg, blk = i - sgm * P.seg_wg_per_seg;
        const bool w = has_work(i);
        const uint32_t at = w ? busy : total + (i - busy);   // the others keep their order behind
        busy += w ? 1u : 0u;
        P.seg_map[at] = make_uint2(sgm, blk);
    }
}

// Down-sampling branch of the converter (TRMSampleRateConverter.m:234-297): one workgroup per voice,
// one thread per output sample.  Output k sits at input time k*inc (16.16, inc > 2^16); both wings walk
// the impulse response at phaseIncrement per tap (:246-270).  The converter's ring index e corresponds
// to tube sample e - pad (TRMRingBuffer.m:36-37); samples outside [0, ntube) are zero.
__global__ __launch_bounds__(256) void trm_downsample_kernel(const Const C, const DownArgs D)
{
    const uint32_t v = blockIdx.x;
    const uint32_t nfr = min(D.nframes[v], D.max_nframes);
    const uint32_t pad = (uint32_t)C.padSize;
    const uint32_t ntube = nfr > 0 ? (nfr - 1) * (uint32_t)C.controlPeriod : 0;
    const uint32_t inc = C.timeRegisterIncrement;
    uint32_t nout = 0;
    if (nfr > 0) nout = (uint32_t)src_count_outputs(ntube, pad, inc);
    const float *x = D.tube + D.tube_offset[v];
    float *out = D.out + D.out_offset[v];
    const int64_t total = (int64_t)ntube + 2 * (int64_t)pad;
    auto sample = [&](int64_t n) {
        n = src_ring_sample(n, total);
        return (n >= 0 && n < (int64_t)ntube) ? x[n] : 0.0f;
    };
    float m = 0.0f;
    for (uint32_t k = threadIdx.x; k < nout; k += blockDim.x) {
        const uint64_t tk = (uint64_t)k * inc;
        const int64_t e = (int64_t)(tk >> 16);
        const uint32_t frac = (uint32_t)(tk & 0xFFFFu);
        float acc = 0.0f;
        uint32_t ph = (uint32_t)__builtin_rint((double)frac * C.sampleRateRatioD);          // :243
        for (int64_t n = e - pad; (ph >> 8) < 3328u; n--, ph += C.phaseIncrement) acc += sample(n) * D.fine[ph];
        ph = (uint32_t)__builtin_rint((double)((~frac) & 0xFFFFu) * C.sampleRateRatioD);     // :257
        for (int64_t n = e + 1 - pad; (ph >> 8) < 3328u; n++, ph += C.phaseIncrement) acc += sample(n) * D.fine[ph];
        out[k] = acc;
        m = fmaxf(m, fabsf(acc));
    }
    __shared__ float sM[256];
    sM[threadIdx.x] = m;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) sM[threadIdx.x] = fmaxf(sM[threadIdx.x], sM[threadIdx.x + s]);
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        D.number_samples[v] = nout;
        D.max_sample[v] = sM[0];
    }
}

// The same branch for the ratios that occur in practice (16 kHz, 8 kHz output from a 19.75 kHz tube), tiled: a
// workgroup = 32 consecutive outputs x 8 voices.  All voices share an output's phase, so its coefficient row (built once
// per batch object, DownArgs::rows) and each voice's window of tube samples are staged in LDS and every thread runs the
// reference's two wing loops over them -- same products in the same order as trm_downsample_kernel (taps past a wing's
// end multiply by 0).  The per-voice maximum is an atomic max on the float's bits (non-negative), max_sample zeroed by
// the launcher.
// Round 4 (measured by ablation, profiles/ab_r04.txt: the kernel is bound by its window loads and its stores, the products
// are a fifth of it): a tile is 64 outputs x 32 voices -- a wave's lanes are 64 consecutive outputs, so a store instruction
// writes 256 contiguous bytes of one voice and the windows overlap their neighbours' by less (1.46 x the samples instead of
// 1.85 x) --, a thread runs one output of EIGHT voices, two taps at a time (one coefficient read feeds eight products), the
// per-voice addresses and counts are fetched before the staging instead of in front of the stores, and the voice's output
// count comes from trm_down_count_kernel (its closed form is four 64-bit divisions: evaluated per thread and voice it was
// most of the kernel's instructions).  Same products in the same order as ever.
constexpr int kDownCols = 64, kDownVoices = 32, kDownPerThread = 8;      // 256 threads: 64 outputs x 4 voice groups x 8 voices each
__global__ __launch_bounds__(kDownCols *kDownVoices / kDownPerThread) void trm_downsample_rows_kernel(const Const C, const DownArgs D, uint32_t xlen)
{
    extern __shared__ float sDown[];
#if defined(TRM_EXP_DOWN) && TRM_EXP_DOWN == 4
    return;
#endif
    const uint32_t T = D.lmax + D.rmax, rp = T | 1u;          // odd LDS pitch: a wave's 64 rows land in different banks
    float *const sRow = sDown;                                // [kDownCols][rp]
    float *const sXw = sDown + kDownCols * rp;                // [kDownVoices][xlen]
    constexpr uint32_t kGroups = kDownVoices / kDownPerThread;
    const uint32_t tid = threadIdx.x, o = tid & (kDownCols - 1), w = tid / kDownCols;      // w: voice group 0..7
    const uint32_t pad = (uint32_t)C.padSize, inc = C.timeRegisterIncrement, CP = (uint32_t)C.controlPeriod;
    // Workgroups are dealt to the 8 XCDs in turn (b and b + 8 share an L2): give every XCD a CONTIGUOUS run of a voice
    // group's time tiles, so that a tile's window overlaps what its own L2 has just fetched and neighbouring PCM lines are
    // written through one L2 (gridDim.x is a multiple of 8: launch_downsample; the surplus tiles return).
    const uint32_t tile = (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    if (tile >= D.ntiles) return;
    const uint32_t k0 = D.k_base + tile * kDownCols, v0 = blockIdx.y * kDownVoices;
    // this thread's voices: their PCM rows and output counts (requested now, used after the products)
    float *outRow[kDownPerThread];
    uint32_t noutOf[kDownPerThread];
    for (int q = 0; q < kDownPerThread; q++) {
        const uint32_t v = v0 + w + q * kGroups;
        const bool ok = v < D.nvoices;
        outRow[q] = D.out + (ok ? D.out_offset[v] : 0ull);
        noutOf[q] = ok ? D.number_samples[v] + D.k_base : 0u;
    }
    // tube samples nOrg <= n < nOrg + nt sit at tube + tube_offset[voice] (one-shot: the voice's own samples from 0)
    const int64_t nOrg = D.stream ? (int64_t)D.n_origin : 0;
    // the block's rows: output k0 + r has phase ((k0 + r) * inc) & 0xFFFF; 8 threads per row
    for (uint32_t r = tid >> 3; r < (uint32_t)kDownCols; r += blockDim.x >> 3) {
        const float *src = D.rows + (size_t)src_phase(k0 + r, inc) * D.pitch;
#if defined(TRM_EXP_DOWN) && TRM_EXP_DOWN == 3
        for (uint32_t i = tid & 7u; i < T; i += 8) sRow[r * rp + i] = 0.5f;
#else
        for (uint32_t i = tid & 7u; i < T; i += 8) sRow[r * rp + i] = src[i];
#endif
    }
    // (no barrier here: the windows are fetched in the same breath as the rows -- every thread reads its voice's offset and
    // length itself instead of through LDS, which cost a memory round trip and a barrier in front of the window loads)
    // every voice's window: tube samples nLo .. nLo + xlen - 1 (zeros outside the voice's own samples); 8 threads per voice.
    // Positions fit 32 bits (utterances end below 2^31 tube samples: trm_capi.cc); past the voice's last sample + flush the
    // converter's ring still holds the lap before (src_ring_sample: down-sampling's "extra lap").
    const int64_t nLo = (int64_t)src_position(k0, inc) - (int64_t)pad - (int64_t)(D.lmax - 1);
    const int32_t nLo32 = (int32_t)(nLo - nOrg);
    for (uint32_t ww = tid >> 3; ww < (uint32_t)kDownVoices; ww += blockDim.x >> 3) {
        const uint32_t vv = v0 + ww;
        const bool okv = vv < D.nvoices;
        const uint32_t nfrv = okv ? min(D.nframes[vv], D.max_nframes) : 0u;
        const float *src = D.tube + (okv ? D.tube_offset[vv] : 0ull);
        const int32_t nt = (int32_t)(D.stream ? (okv ? (uint32_t)(D.n_hi - D.n_origin) : 0u) : (nfrv > 0 ? (nfrv - 1) * CP : 0u));
        const int32_t total = nt + 2 * (int32_t)pad;
        for (uint32_t i = tid & 7u; i < xlen; i += 8) {
            int32_t n = nLo32 + (int32_t)i;
            if (!D.stream && n >= total) n -= (int32_t)kSrcRing * ((n - total) / (int32_t)kSrcRing + 1);      // (one-shot: nt = the voice's tube samples)
#if defined(TRM_EXP_DOWN) && TRM_EXP_DOWN == 2
            sXw[ww * xlen + i] = (n >= 0 && n < nt) ? 0.25f : 0.0f;
#else
            sXw[ww * xlen + i] = (n >= 0 && n < nt) ? src[n] : 0.0f;
#endif
        }
    }
    __syncthreads();
    const uint32_t k = k0 + o;
    const float *row = &sRow[o * rp];
    const uint32_t centre = (uint32_t)((int64_t)src_position(k, inc) - (int64_t)pad - nLo);       // tube sample e - pad in the window
    // this thread's voices: w, w + kGroups, ... of the tile; one coefficient read feeds a product of each
    float acc[kDownPerThread];
    const float *xw[kDownPerThread];
    for (int q = 0; q < kDownPerThread; q++) {
        acc[q] = 0.0f;
        xw[q] = &sXw[(w + q * kGroups) * xlen + centre];
    }
#if defined(TRM_EXP_DOWN) && TRM_EXP_DOWN == 1
    for (int q = 0; q < kDownPerThread; q++) acc[q] = xw[q][0] * row[q];
    if (false)
#endif
    {
        uint32_t j = 0;
        for (; j + 1 < D.lmax; j += 2) {
            const float c0 = row[j], c1 = row[j + 1];
            for (int q = 0; q < kDownPerThread; q++) {
                const float x0 = xw[q][-(int)j], x1 = xw[q][-(int)j - 1];
                acc[q] += x0 * c0;
                acc[q] += x1 * c1;
            }
        }
        if (j < D.lmax) {
            const float c = row[j];
            for (int q = 0; q < kDownPerThread; q++) acc[q] += xw[q][-(int)j] * c;
        }
        const float *rrow = row + D.lmax;
        for (j = 0; j + 1 < D.rmax; j += 2) {
            const float c0 = rrow[j], c1 = rrow[j + 1];
            for (int q = 0; q < kDownPerThread; q++) {
                const float x0 = xw[q][1 + j], x1 = xw[q][2 + j];
                acc[q] += x0 * c0;
                acc[q] += x1 * c1;
            }
        }
        if (j < D.rmax) {
            const float c = rrow[j];
            for (int q = 0; q < kDownPerThread; q++) acc[q] += xw[q][1 + j] * c;
        }
    }
    for (int q = 0; q < kDownPerThread; q++) {
        const uint32_t wl = w + q * kGroups, v = v0 + wl;
        // (the voice's output count was put in number_samples by trm_down_count_kernel: the closed form is four 64-bit
        // divisions, and evaluated here -- per thread and voice, as rounds 1-3 did -- it cost ten times the products)
        const uint32_t nout = noutOf[q];
        float m = 0.0f;
#if defined(TRM_EXP_DOWN) && TRM_EXP_DOWN == 5
        if (k < nout && acc[q] == 12345.678f) {
#else
        if (k < nout) {
#endif
            outRow[q][k - D.k_base] = acc[q];
            m = fabsf(acc[q]);
        }
        for (int off = kDownCols / 2; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, kWave));     // the lanes of this voice
        if (o == 0 && v < D.nvoices) {
            if (m > 0.0f) atomicMax(reinterpret_cast<unsigned int *>(&D.max_sample[v]), __float_as_uint(m));
        }
    }
}

// number_samples[v] of a down-sampling launch (what trm_downsample_rows_kernel's workgroups mask their stores with):
// the outputs of a voice of (nframes - 1) * controlPeriod tube samples (src_count_outputs), a chunk's k_end - k_base
__global__ __launch_bounds__(256) void trm_down_count_kernel(const Const C, const DownArgs D)
{
    const uint32_t v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= D.nvoices) return;
    const uint32_t nfr = min(D.nframes[v], D.max_nframes);
    uint32_t nout = nfr > 0 ? (uint32_t)src_count_outputs((uint64_t)(nfr - 1) * (uint32_t)C.controlPeriod, (uint32_t)C.padSize, C.timeRegisterIncrement) : 0u;
    if (D.stream) nout = D.k_end;
    D.number_samples[v] = nout - D.k_base;
}

// Output normalisation (TRMTubeModel.m:370-389 file path, :515-533 WAV-data path).  One
// workgroup per voice; mono -> int16[n], stereo -> interleaved int16[2n].
__global__ __launch_bounds__(256) void trm_int16_kernel(const ScaleArgs S)
{
    const uint32_t v = blockIdx.x;
    const uint32_t n = S.number_samples[v];
    const float mx = S.max_sample[v];
    const float *src = S.pcm + S.out_offset[v];
    const double scale = (32767.0 / (double)mx) * S.volumeAmp;
    if (S.channels == 2) {
        const double g = S.forWavData ? 1.0 : 2.0;
        const double left = -((S.balance / 2.0) - 0.5) * scale * g;
        const double right = ((S.balance / 2.0) + 0.5) * scale * g;
        int16_t *dst = S.pcm16 + 2 * S.out_offset[v];
        for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) {
            double x = (double)src[i];
            dst[2 * i] = (int16_t)(uint16_t)(int64_t)__builtin_rint(x * left);        // wraps like the reference
            dst[2 * i + 1] = (int16_t)(uint16_t)(int64_t)__builtin_rint(x * right);
        }
    } else {
        int16_t *dst = S.pcm16 + S.out_offset[v];
        for (uint32_t i = threadIdx.x; i < n; i += blockDim.x)
            dst[i] = (int16_t)(uint16_t)(int64_t)__builtin_rint((double)src[i] * scale);
    }
}

// Sound-file images on the device: -saveOutputToFile:error: (TRMTubeModel.m:365-490) for every voice of a batch --
// the container's header (AU 24 bytes, AIFF 54, WAVE 44; the byte layouts of csrc/trm_io.cc's host writer) followed by the
// int16 payload in the container's byte order (AU / AIFF big-endian, WAVE little-endian, :410-412), scaled like
// trm_int16_kernel's file form (x2 stereo gains, :382-383) -- at files + file_offset[v]: one D2H, or a write() straight from a
// mapped buffer, gives ready files.  One workgroup per voice.
__global__ __launch_bounds__(256) void trm_file_image_kernel(const FileArgs F)
{
    const uint32_t v = blockIdx.x;
    const uint32_t n = F.s.number_samples[v];
    const float mx = F.s.max_sample[v];
    const float *src = F.s.pcm + F.s.out_offset[v];
    uint8_t *img = F.files + F.file_offset[v];
    const uint32_t ch = F.s.channels == 2 ? 2u : 1u, bytes = n * ch * 2u;
    const uint32_t hdr = F.format == 0 ? 24u : F.format == 1 ? 54u : 44u;
    if (threadIdx.x < hdr) {
        // header byte i: fixed bytes from the host's template (magic words, rate, channel count, the AIFF rate as an 80-bit
        // extended), the size fields filled in here (they depend on the voice's sample count)
        const uint32_t i = threadIdx.x;
        uint8_t b = F.header[i];
        auto be = [&](uint32_t at, uint32_t val) { if (i >= at && i < at + 4) b = (uint8_t)(val >> (8 * (3 - (i - at)))); };
        auto le = [&](uint32_t at, uint32_t val) { if (i >= at && i < at + 4) b = (uint8_t)(val >> (8 * (i - at))); };
        if (F.format == 0) be(8, bytes);
        else if (F.format == 1) { be(4, 4 + 8 + 18 + 8 + 8 + bytes); be(22, n); be(42, 8 + bytes); }
        else { le(4, 36 + bytes); le(40, bytes); }
        img[i] = b;
    }
    const double scale = (32767.0 / (double)mx) * F.s.volumeAmp;
    const double left = ch == 2 ? -((F.s.balance / 2.0) - 0.5) * scale * 2.0 : scale;
    const double right = ((F.s.balance / 2.0) + 0.5) * scale * 2.0;
    const bool big = F.format != 2;
    uint8_t *body = img + hdr;
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) {
        const double x = (double)src[i];
        uint16_t a = (uint16_t)(int64_t)__builtin_rint(x * left);              // wraps like the reference's x86 cast
        if (big) a = (uint16_t)((a << 8) | (a >> 8));
        if (ch == 2) {
            uint16_t r = (uint16_t)(int64_t)__builtin_rint(x * right);
            if (big) r = (uint16_t)((r << 8) | (r >> 8));
            body[4 * i] = (uint8_t)a; body[4 * i + 1] = (uint8_t)(a >> 8); body[4 * i + 2] = (uint8_t)r; body[4 * i + 3] = (uint8_t)(r >> 8);
        } else {
            body[2 * i] = (uint8_t)a; body[2 * i + 1] = (uint8_t)(a >> 8);
        }
    }
}

// out[v * pitch + i] *= g for i < count; mx[v] *= g  (streams in TRAcT's loop order: tube.c:1177's x100)
__global__ __launch_bounds__(256) void trm_gain_kernel(float *out, size_t pitch, uint32_t count, uint32_t nvoices, float *mx, float g)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    // (voices on a stride of gridDim.y: a grid's y extent ends at 65535, wide streams carry millions of voices)
    for (uint32_t v = blockIdx.y; v < nvoices; v += gridDim.y) {
        float *row = out + (size_t)v * pitch;
        if (i < count) row[i] *= g;
        if (i == 0 && mx) mx[v] *= g;
    }
}

// ---------------------------------------------------------------- launchers (host)
hipError_t launch_noise(float *lp, uint32_t from, uint32_t to, double *state, hipStream_t stream)
{
    hipLaunchKernelGGL(trm_noise_kernel, dim3(1), dim3(1), 0, stream, lp, from, to, state);
    return hipGetLastError();
}

hipError_t launch_tube(const Const &c, const TubeArgs &a, hipStream_t stream)
{
    if (a.nvoices == 0) return hipSuccess;
    // (time-split: one workgroup per segment and block of 64 voices; seg_wg_per_seg * segments, set by the caller in wg_base's
    // place holder `seg_grid`)
    const uint32_t grid = a.seg_periods ? a.seg_grid : (a.nvoices + kWave - 1) / kWave;
    // A grid of more than two rounds of resident workgroups (2 per CU) runs measurably slower per workgroup than its first
    // two rounds (MI355X, 256 CUs: 1024 workgroups 18.1 ms, 1536: 32.6, 2048: 40.6 -- profiles/ab_r03.txt): the
    // batch goes out in slices of `slice` workgroups, back to back on the stream.  TRM_WIDE_SLICE overrides (0 = one launch).
    static const uint32_t slice = [] {
        const char *e = getenv("TRM_WIDE_SLICE");
        return e ? (uint32_t)strtoul(e, nullptr, 10) : 1024u;
    }();
    TubeArgs s = a;
    for (uint32_t base = 0; base < grid;) {
        const uint32_t n = slice == 0 ? grid - base : (grid - base < slice ? grid - base : slice);
        s.wg_base = base;
        if (a.seg_periods) hipLaunchKernelGGL(trm_tube_kernel<kModeSegments>, dim3(n), dim3(kWave * kRoles), 0, stream, c, s);
        else if (a.stream_state) hipLaunchKernelGGL(trm_tube_kernel<kModeStream>, dim3(n), dim3(kWave * kRoles), 0, stream, c, s);
        else hipLaunchKernelGGL(trm_tube_kernel<kModeOneShot>, dim3(n), dim3(kWave * kRoles), 0, stream, c, s);
        base += n;
    }
    return hipGetLastError();
}

hipError_t launch_phase(const Const &c, const PhaseArgs &a, hipStream_t stream)
{
    if (a.nvoices == 0 || a.nseg == 0) return hipSuccess;
    const uint64_t periods = (uint64_t)a.nvoices * a.max_nframes;
    hipLaunchKernelGGL(trm_phase_period_kernel, dim3((unsigned)((periods + 255) / 256)), dim3(256), 0, stream, c, a);
    const uint64_t threads = (uint64_t)(a.nseg - 1) * a.seg_wg_per_seg * a.voices_per_wg;
    if (threads > 0) hipLaunchKernelGGL(trm_phase_segment_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, stream, c, a);
    if (a.seg_map && a.block_frames) {
        hipLaunchKernelGGL(trm_seg_blocks_kernel, dim3((a.seg_wg_per_seg * a.voices_per_wg + 255) / 256), dim3(256), 0, stream, a);
        hipLaunchKernelGGL(trm_seg_map_kernel, dim3(1), dim3(kMapThreads), 0, stream, a);
    }
    return hipGetLastError();
}

// resident workgroups of trm_tube_kernel per CU as the runtime computes it (diagnostics / DESIGN.md)
int tube_kernel_blocks_per_cu()
{
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, trm_tube_kernel<kModeOneShot>, kWave * kRoles, 0) != hipSuccess) return -1;
    return n;
}

// LDS of the tiled down-sampling kernel for a converter of lmax + rmax taps: 64 coefficient rows + 32 voices' windows
static size_t down_tile_lds(const Const &c, uint32_t lmax, uint32_t rmax, uint32_t *xlenOut)
{
    const uint32_t T = lmax + rmax;
    const uint32_t xlen = (uint32_t)(((uint64_t)(kDownCols - 1) * c.timeRegisterIncrement) >> 16) + 2u + T;
    if (xlenOut) *xlenOut = xlen;
    return ((size_t)kDownCols * (T | 1u) + (size_t)kDownVoices * xlen) * sizeof(float);
}
// up to 48 KB (three workgroups per CU) at speech-rate ratios; the widest converter a stream accepts (output at a quarter of the
// tube rate) needs 73 KB
constexpr size_t kDownLdsMax = 96 * 1024;
bool downsample_tiled_fits(const Const &c, uint32_t lmax, uint32_t rmax)
{
    return lmax + rmax > 0 && down_tile_lds(c, lmax, rmax, nullptr) <= kDownLdsMax;
}

hipError_t launch_downsample(const Const &c, const DownArgs &a, hipStream_t stream)
{
    if (a.nvoices == 0) return hipSuccess;
    // the tiled kernel where its rows and windows fit LDS (any ratio a stream accepts does), else the generic walk
    uint32_t xlen = 0;
    const size_t lds = down_tile_lds(c, a.lmax, a.rmax, &xlen);
    if (a.rows && downsample_tiled_fits(c, a.lmax, a.rmax)) {
        if (lds > 48 * 1024) {
            static DynamicLdsAllowance allowance;
            hipError_t ea = allowance.ensure(reinterpret_cast<const void *>(trm_downsample_rows_kernel), (int)kDownLdsMax);
            if (ea != hipSuccess) return ea;
        }
        const uint32_t ntubeMax = a.max_nframes > 0 ? (a.max_nframes - 1) * (uint32_t)c.controlPeriod : 0;
        // (a shorter voice may end on the reference's extra lap, src_count_outputs: cover it)
        uint64_t noutMax = (((uint64_t)ntubeMax + 2ull * (uint32_t)c.padSize + kSrcRing) * 65536ull + c.timeRegisterIncrement - 1) / c.timeRegisterIncrement;
        if (a.stream) noutMax = a.k_end - a.k_base;
        hipError_t e = hipMemsetAsync(a.max_sample, 0, a.nvoices * sizeof(float), stream);
        if (e != hipSuccess) return e;
        DownArgs t = a;
        t.ntiles = (uint32_t)((noutMax + kDownCols - 1) / kDownCols > 0 ? (noutMax + kDownCols - 1) / kDownCols : 1);
        const dim3 grid((t.ntiles + 7u) & ~7u, (a.nvoices + kDownVoices - 1) / kDownVoices);
        hipLaunchKernelGGL(trm_down_count_kernel, dim3((a.nvoices + 255) / 256), dim3(256), 0, stream, c, t);
        hipLaunchKernelGGL(trm_downsample_rows_kernel, grid, dim3(kDownCols * kDownVoices / kDownPerThread), lds, stream, c, t, xlen);
        return hipGetLastError();
    }
    if (a.stream) return hipErrorInvalidValue;       // (the generic kernel converts whole utterances only)
    hipLaunchKernelGGL(trm_downsample_kernel, dim3(a.nvoices), dim3(256), 0, stream, c, a);
    return hipGetLastError();
}

hipError_t launch_gain(float *out, size_t pitch, uint32_t count, uint32_t nvoices, float *mx, float g, hipStream_t stream)
{
    if (nvoices == 0 || count == 0) return hipSuccess;
    hipLaunchKernelGGL(trm_gain_kernel, dim3((count + 255) / 256, nvoices < 32768u ? nvoices : 32768u), dim3(256), 0, stream, out, pitch, count,
                       nvoices, mx, g);
    return hipGetLastError();
}

hipError_t launch_file_images(const FileArgs &f, uint32_t nvoices, hipStream_t stream)
{
    if (nvoices == 0) return hipSuccess;
    hipLaunchKernelGGL(trm_file_image_kernel, dim3(nvoices), dim3(256), 0, stream, f);
    return hipGetLastError();
}

hipError_t launch_int16(const ScaleArgs &s, uint32_t nvoices, hipStream_t stream)
{
    if (nvoices == 0) return hipSuccess;
    hipLaunchKernelGGL(trm_int16_kernel, dim3(nvoices), dim3(256), 0, stream, s);
    return hipGetLastError();
}

}  // namespace trm
