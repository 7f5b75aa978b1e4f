// trm_oct.hip -- trm_tube_kernel_o: -[TRMTubeModel synthesize] (TRMTubeModel.m:272-361) with EIGHT lanes per voice and
// 8 voices per workgroup, for one-shot batches of at most two workgroups per CU (4096 voices on 256 CUs).
//
// trm_quad.hip's workgroup (16 voices, four lanes each, six waves) leaves every wave with ~40-50 instructions per tube
// sample, and a wave issues one instruction per ~4.85 cycles whatever it is (profiles/valu_ceiling_r02.txt): each role
// is its own serial floor of ~2.4-2.6 ms per second of speech, whichever SIMD it sits on.  Here the same six roles carry
// HALF the voices per pass, so that two workgroups share a CU (three waves per SIMD fill each other's issue slots):
//   osc, mix, area, fric   lanes = 8 consecutive tube samples of a voice: lanes 0-31 are the step's first block of four
//                          (rows of 16 lanes = 4 slots x 4 voices, exactly trm_quad.hip's layout), lanes 32-63 its second
//                          block.  The recurrences that run over slots -- oscillator phase (a prefix sum), throat low-pass
//                          and frication band-pass (serial scans) -- cross the halves with v_permlane32_swap.
//   tube                   lanes = 8 parts of the tube (trm_oct.h): ~35 instructions per sample instead of 50
//   convert                lane = output time (rows of 32 outputs x 2 voices), two row pairs per block
// One barrier per step of 8 tube samples.  At step i osc works on block i, mix on i-1, the coefficient waves and the two
// scans on i-2, tube on i-4, convert on whatever is complete, metered.  No streaming instance (trm_quad.hip carries streams).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "trm_devutil.h"
#include "trm_kernels.h"
#include "trm_lane.h"
#include "trm_oct.h"
#include "trm_quad.h"
#include "trm_quad_dev.h"

#ifndef TRM_EXPERIMENTS
#undef TRM_OCT_ROLE_PERM
#undef TRM_OCT_TUBE_PRIO
#undef TRM_OCT_PRIO_OSC
#undef TRM_OCT_PRIO_MIX
#undef TRM_OCT_PRIO_COEF
#undef TRM_OCT_PRIO_COEF2
#undef TRM_OCT_PRIO_CVT
#undef TRM_ABL_SKIP
#undef TRM_ISA_ROLE
#endif

namespace trm {

constexpr int kOV = 8;               // voices per workgroup
constexpr int kOB = 8;               // tube samples per step = time slots per voice (two blocks of kSlots)
constexpr int kORoles = 6;           // osc, mix, coef x2 (area | frication), tube, convert
// mix -> tube: the per-voice values of a sample as FOUR arrays [buffer][slot][voice] of floats -- glottal input, noise
// signal, throat input / output, the three-way junction's alpha -- so that a wave's 32-lane groups touch 32 consecutive
// dwords (one float4 record per voice put 32 lanes on 8 banks); the arrays start 8 banks apart: the tube wave's parts read
// three of them in one instruction
constexpr int kOXArray = kXDepth * kOB * kOV + 8;
enum { kXGin = 0, kXSig = 1, kXThr = 2, kXAlpha = 3 };
// coef -> tube: one float4 record {k.x, k.y | in.x, in.y} per (sample, part, voice).  ds_read_b128 serves a wave in four
// groups of 16 lanes -- {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} and the same + 32 (MI355X_MICROARCH.md, LDS) -- which for
// the tube wave's lanes (voice v = lane / 8, part p = lane % 8) are four blocks {4 parts of one voice}: this index gives
// every lane of a group its own 16-byte bank slot 4 (p & 3) + (v & 3)  (the first layout, part rows of 8 + 2 records,
// was 2-way conflicted there: SQ_LDS_BANK_CONFLICT 1.4e8 -> see profiles/)
__host__ __device__ constexpr int oct_k_index(int p, int v) { return 16 * ((p >> 2) * 2 + (v >> 2)) + 4 * (p & 3) + (v & 3); }
// Control frames, staged in LDS ahead of their use by LDS-DMA (the area wave, which has no other vector-memory traffic,
// issues one 512-byte transfer per control period: 8 voices x 64 bytes) and read by the three waves that turn them into
// tracks when a period starts.  Frame f lives in slot f % 8.  When the oscillator wave enters the period between frames
// p and p+1, the stager retires the transfer of frame p+2 (sent for a whole period earlier: the wait is free) and sends
// for frame p+3; frames p-1 .. p+1 are what the lagging waves may still be reading.  Carrying the prefetched frames in
// REGISTERS instead (r01, r02) cost every coefficient wave 26 register-to-register copies per step -- the compiler's way
// of keeping a conditionally rotated array across the loop's back edge -- a fifth of their instructions.
constexpr int kFrameRing = 8;
constexpr int kOKRow = 64 + 4;       // float4s per (buffer, sample): 64 records + 64 bytes, so that the writers' time slots
                                     // alternate between the two halves of the 32 write banks

// LDS of one workgroup, one dynamically sized array (see trm_quad.hip: a static size makes the compiler pad the VGPR
// allocation for "three waves per SIMD", and two co-resident workgroups then do not fit)
struct OctLds {
    static constexpr size_t oO = 0;                                                       // float2 [kOV * kOStride]
    static constexpr size_t oA = oO + sizeof(float2) * kOV * kOStride;                    // float2 [2 * kWave]
    static constexpr size_t oX = oA + sizeof(float2) * 2 * kWave;                         // float  [4 * kOXArray]
    static constexpr size_t oK = oX + sizeof(float) * 4 * kOXArray;                       // float4 [kKDepth * kOB * kOKRow]
    static constexpr size_t oY = oK + sizeof(float4) * kKDepth * kOB * kOKRow;            // float  [kOV * kYStride]
    static constexpr size_t oRows = oY + sizeof(float) * kOV * kYStride;                  // float  [kRowBufs * kCvtCols * kRowPitch]
    static constexpr size_t oInfo = oRows + sizeof(float) * kRowBufs * kCvtCols * kRowPitch;   // uint4 [kOV]
    static constexpr size_t oMx = oInfo + sizeof(uint4) * kOV;                            // float  [4 * kWave]
    static constexpr size_t oNoise = oMx + sizeof(float) * 4 * kWave;                     // float  [kNoiseRing]
    static constexpr size_t oSync = oNoise + sizeof(float) * kNoiseRing;                  // uint32 [2] (+ 8 bytes)
    static constexpr size_t oFrames = oSync + 16;                                         // float4 [kFrameRing * kOV * 4]
    static constexpr size_t kBytes = oFrames + sizeof(float4) * kFrameRing * kOV * 4;
    static_assert(oA % 16 == 0 && oX % 16 == 0 && oK % 16 == 0 && oY % 16 == 0 && oRows % 16 == 0 && oInfo % 16 == 0, "16-byte aligned pieces");
    static_assert(2 * kBytes <= 160 * 1024, "two workgroups per CU");
};

// every lane's value as lanes 0-31 hold it / as lanes 32-63 hold it (lane l and l + 32 see the same pair)
struct HalfPair { float lo, hi; };
struct HalfPairU { unsigned lo, hi; };
__device__ __forceinline__ HalfPairU across_halves(unsigned u)
{
    const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return HalfPairU{r[0], r[1]};
}
__device__ __forceinline__ HalfPair across_halves(float x)
{
    const HalfPairU r = across_halves(__builtin_bit_cast(unsigned, x));
    return HalfPair{__builtin_bit_cast(float, r.lo), __builtin_bit_cast(float, r.hi)};
}

__global__ __launch_bounds__(kWave *kORoles, 4) void trm_tube_kernel_o(const Const C, const TubeArgs A)
{
    typedef OctLds L;
    extern __shared__ __attribute__((aligned(16))) unsigned char sLds[];
    float2 *const sO = reinterpret_cast<float2 *>(sLds + L::oO);           // osc -> mix: oscillator reads
    float2 *const sA = reinterpret_cast<float2 *>(sLds + L::oA);           // osc -> mix: {ax, ah1} per (step & 1, lane)
    float *const sX = reinterpret_cast<float *>(sLds + L::oX);             // mix -> tube: [gin | sig | thr | alpha][buf][slot][voice]
    float4 *const sK = reinterpret_cast<float4 *>(sLds + L::oK);           // coef -> tube: {k | injections} [buf][slot][part][voice]
    float *const sY = reinterpret_cast<float *>(sLds + L::oY);             // tube-rate rings
    float *const sRows = reinterpret_cast<float *>(sLds + L::oRows);       // mix -> convert: coefficient rows of 3 blocks
    uint4 *const sInfo = reinterpret_cast<uint4 *>(sLds + L::oInfo);
    float *const sMx = reinterpret_cast<float *>(sLds + L::oMx);
    float *const sNoise = reinterpret_cast<float *>(sLds + L::oNoise);
    uint32_t *const sRowSync = reinterpret_cast<uint32_t *>(sLds + L::oSync);   // [0] convert -> mix: first block whose staged rows are still needed; [1] mix -> convert: blocks staged
    float4 *const sF = reinterpret_cast<float4 *>(sLds + L::oFrames);          // control frames [f % kFrameRing][voice][quarter]

    constexpr int kStampRoles = kORoles;
    (void)kStampRoles;
    constexpr int kThreads = kWave * kORoles;
    const int lane = threadIdx.x & (kWave - 1);
    const int waveIdx = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
#ifndef TRM_OCT_ROLE_PERM
#define TRM_OCT_ROLE_PERM 0, 1, 3, 4, 2, 5      /* osc mix fric tube | area convert: waves w and w+4 of a workgroup share a SIMD, the
                                                   frication and tube waves have one to themselves (and the other workgroup's
                                                   waves for company); of 20 assignments tried this is the fastest by 1-10 %
                                                   (profiles/oct_variants_r02.txt) */
#endif
    const int rolePerm[kORoles] = {TRM_OCT_ROLE_PERM};
    int role = 0;
    for (int i = 0; i < kORoles; i++) role = waveIdx == i ? rolePerm[i] : role;
#ifdef TRM_ISA_ROLE      // instruction-count studies (tools/isa_mix.py role N): every wave is this role, the other bodies fold away
    role = TRM_ISA_ROLE;
#endif

    // feed-forward waves: lane -> (voice, slot); a row of 16 lanes = 4 banks (slot within the block) x 4 voices, lanes
    // 0-31 the step's first block, 32-63 its second.  Tube wave: lane -> (voice, part), 8 consecutive lanes per voice.
    const bool upperHalf = lane >= 32;
    const int part = (lane >> 2) & 3;                                    // slot within the block
    const int slot = (upperHalf ? 4 : 0) + part;                         // slot within the step
    const int vq = role == 4 ? lane >> 3 : ((lane >> 4) & 1) * 4 + (lane & 3);      // voice within the workgroup
    const uint32_t vRaw = blockIdx.x * kOV + vq;
    const bool laneValid = vRaw < A.nvoices;
    const uint32_t v = laneValid ? vRaw : A.nvoices - 1;

    const uint32_t nfr = min(A.nframes[v], A.max_nframes);
    const uint32_t nfrMax = wave_max_u32(nfr);
    const uint32_t CP = (uint32_t)C.controlPeriod;
    const uint32_t inc = C.timeRegisterIncrement;
    const uint32_t ntubeMax = nfrMax > 0 ? (nfrMax - 1) * CP : 0;
    // tube samples the tube stage produces: the utterance, then the converter's 2*pad zero flush (TRMRingBuffer.m:85-93)
    const uint32_t nTotal = nfrMax > 0 ? ntubeMax + 2u * (uint32_t)C.padSize : 0;
    // the tube stage steps block i-4 at step i; the convert wave finishes what is queued after the last barrier
    const uint32_t nSteps = nTotal > 0 ? (nTotal + kOB - 1) / kOB + 5 : 0;
    const uint32_t ntubeLane = nfr > 0 ? (nfr - 1) * CP : 0;
    const uint32_t ntubeMin = wave_min_u32(ntubeLane);      // every voice of the group is still sounding below this
    // where sample `s` of block `blk` keeps this lane's voice's value of array `arr`
    auto x_at = [&](int arr, uint32_t blk, int s) -> float & { return sX[arr * kOXArray + ((blk % kXDepth) * kOB + s) * kOV + vq]; };

    // this lane's voice's frame f (nominal index: past the voice's last frame the stager repeats it), quarter q
    auto frame_q = [&](uint32_t f, int q) { return sF[((f % kFrameRing) * kOV + vq) * 4 + q]; };
    auto frame_to = [&](uint32_t f, float *dst, int quads) {
        for (int q = 0; q < quads; q++) {
            const float4 x = frame_q(f, q);
            dst[4 * q] = x.x; dst[4 * q + 1] = x.y; dst[4 * q + 2] = x.z; dst[4 * q + 3] = x.w;
        }
    };
    // the stager's lanes (area wave, lanes 0-31): lane -> (voice lane / 4, quarter lane % 4) of the workgroup's 8 frames
    const float *stageSrc = nullptr;
    uint32_t stageNfr = 0;
    if (role == 2) {
        const uint32_t sv = min(blockIdx.x * kOV + ((uint32_t)lane >> 2 & 7u), A.nvoices - 1);
        stageNfr = min(A.nframes[sv], A.max_nframes);
        stageSrc = A.frames + (stageNfr > 0 ? A.frame_offset[sv] * 16 : 0) + (lane & 3) * 4;
    }
    auto stage_frame = [&](uint32_t f) {
        const uint32_t fi = stageNfr > 0 ? (f < stageNfr ? f : stageNfr - 1) : 0u;
        if (lane < 32) dma16(stageSrc + (size_t)fi * 16, reinterpret_cast<float *>(&sF[(f % kFrameRing) * kOV * 4]));
    };

    for (int i = threadIdx.x; i < kOV * kYStride; i += kThreads) sY[i] = 0.0f;
    for (int i = threadIdx.x; i < kOV * kOStride; i += kThreads) sO[i] = make_float2(0.0f, 0.0f);
    for (int i = threadIdx.x; i < kKDepth * kOB * kOKRow; i += kThreads) sK[i] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    for (int i = threadIdx.x; i < 4 * kOXArray; i += kThreads) sX[i] = 0.0f;
    if (threadIdx.x < 2) sRowSync[threadIdx.x] = 0u;
    __syncthreads();
    // what no wave writes per sample: the nasal tract's constant records (parts 6, 7; part 5's injections stay 0)
    for (int i = threadIdx.x; i < kKDepth * kOB * kOV; i += kThreads) {
        float4 *row = &sK[(i / kOV) * kOKRow];
        row[oct_k_index(6, i % kOV)] = make_float4(C.nasalTd[1], C.nasalTd[2], 0.0f, 0.0f);
        row[oct_k_index(7, i % kOV)] = make_float4(C.nasalTd[3], C.nasalK6a, 0.0f, C.onePlusNK6);
    }
    if (role == 2 && nSteps > 0) {
        stage_frame(0); stage_frame(1); stage_frame(2); stage_frame(3);
        dma_wait_all();
    }
    __syncthreads();

    // The two recurrences that only FEED the tube (frication band-pass, throat low-pass) run in feed-forward waves,
    // serially over a voice's slots: slot s's output is slot s+1's y1 and slot s+2's y2.  A pass of four turns settles
    // the four slots of a block (a lane's evaluation is final from its own turn on); the step's two blocks are two passes:
    // the first settles lanes 0-31, its end state crosses to lanes 32-63, the second settles those, and their end state
    // crosses back for the next step.
    struct ScanState { float by1, by2, ny1, ny2, prevSig, thY, thNext; };
    auto scans_reset = [&](ScanState &Z) { Z.by1 = Z.by2 = Z.ny1 = Z.ny2 = Z.prevSig = Z.thY = Z.thNext = 0.0f; };
    // band-pass (TRMFilters.m:19-29) of one block of four: the lane's input `sig`, coefficients bp = {2 alpha, 2 beta, 2 gamma}
    auto bandpass_pass = [&](ScanState &Z, float sig, const float4 bp) {
        const float X = q_take<0, kPart2 | kPart3>(sig, Z.prevSig);         // slots 0, 1 look into the previous block
        const float x2 = q_take<2, kPartAll>(X, X);
        float f = 0.0f;
#pragma unroll
        for (int t = 0; t < kSlots; t++) {
            f = bandpass_eval<float>(bp.x, bp.y, bp.z, sig, x2, Z.by1, Z.by2);
            if (t == 0) { Z.by1 = q_take<1, kPart1>(Z.by1, f); Z.by2 = q_take<2, kPart2>(Z.by2, f); }
            if (t == 1) { Z.by1 = q_take<1, kPart2>(Z.by1, f); Z.by2 = q_take<2, kPart3>(Z.by2, f); }
            if (t == 2) { Z.by1 = q_take<1, kPart3>(Z.by1, f); Z.ny2 = q_take<2, kPart0>(Z.ny2, f); }
            if (t == 3) { Z.ny1 = q_take<1, kPart0>(Z.ny1, f); Z.ny2 = q_take<2, kPart1>(Z.ny2, f); }
        }
        Z.by1 = q_take<0, kPart0>(Z.by1, Z.ny1);
        Z.by2 = q_take<0, kPart0 | kPart1>(Z.by2, Z.ny2);
        Z.prevSig = sig;
        return f;
    };
    auto bandpass_scan = [&](ScanState &Z, float sig, const float4 bp) {
        const float fLo = bandpass_pass(Z, sig, bp);            // final in lanes 0-31
        {   // the first block's end state -> the second block's lanes
            const HalfPair a = across_halves(Z.by1), b = across_halves(Z.by2), c = across_halves(Z.prevSig);
            Z.by1 = upperHalf ? a.lo : Z.by1;
            Z.by2 = upperHalf ? b.lo : Z.by2;
            Z.prevSig = upperHalf ? c.lo : Z.prevSig;
        }
        const float fHi = bandpass_pass(Z, sig, bp);            // final in lanes 32-63
        {   // the second block's end state -> the first block's lanes, for the next step
            const HalfPair a = across_halves(Z.by1), b = across_halves(Z.by2), c = across_halves(Z.prevSig);
            Z.by1 = upperHalf ? Z.by1 : a.hi;
            Z.by2 = upperHalf ? Z.by2 : b.hi;
            Z.prevSig = upperHalf ? Z.prevSig : c.hi;
        }
        return upperHalf ? fHi : fLo;
    };
    // throat low-pass (:341, TRMFilters.m:72-77) over this lane's input `thr`
    auto throat_pass = [&](ScanState &Z, float thr) {
        float ty = 0.0f;
#pragma unroll
        for (int t = 0; t < kSlots; t++) {
            ty = throat_eval<float>(C, thr, Z.thY);
            if (t == 0) Z.thY = q_take<1, kPart1>(Z.thY, ty);
            if (t == 1) Z.thY = q_take<1, kPart2>(Z.thY, ty);
            if (t == 2) Z.thY = q_take<1, kPart3>(Z.thY, ty);
            if (t == 3) Z.thNext = q_take<1, kPart0>(Z.thNext, ty);
        }
        Z.thY = q_take<0, kPart0>(Z.thY, Z.thNext);
        return ty;
    };
    auto throat_scan = [&](ScanState &Z, float thr) {
        const float tLo = throat_pass(Z, thr);
        { const HalfPair a = across_halves(Z.thY); Z.thY = upperHalf ? a.lo : Z.thY; }
        const float tHi = throat_pass(Z, thr);
        { const HalfPair a = across_halves(Z.thY); Z.thY = upperHalf ? Z.thY : a.hi; }
        return upperHalf ? tHi : tLo;
    };

#ifdef TRM_ABL_SKIP      // timing experiments only: the masked roles keep the barriers and do nothing
    if ((TRM_ABL_SKIP >> role) & 1) {
        for (uint32_t step = 0; step < nSteps; step++) step_barrier();
        return;
    }
#endif
    if (role == 0) {
#ifndef TRM_OCT_PRIO_OSC
#define TRM_OCT_PRIO_OSC 1
#endif
        __builtin_amdgcn_s_setprio(TRM_OCT_PRIO_OSC);
        // ------------------------------------------------------------ osc: block i at step i, lane = (voice, slot)
        auto sine = [&](int i) { return sine_table(i); };
        // The control-period set-up (four fp64 exponentials) runs ONCE per period and wave: the lanes of a voice enter a
        // period in two different steps (its first sample falls somewhere inside a step), so the step in which the first of
        // them does sets the period's track up for every lane at the lane's own entry position (Tn), and a lane adopts it
        // when its sample crosses (a control period holds at least two steps: launch_tube_oct).  Each exponential is
        // evaluated in one of a voice's four slots of the block and handed to the other three.
        OscSlotTrack T, Tn;
        double P = 0.0;                                 // oscillator position at the start of the step
        uint32_t perN = 1, bnd = CP;                    // the period after the one last set up runs between frames perN, perN + 1; bnd: its first sample
        uint32_t j = (uint32_t)slot;                    // position of this lane's sample in its control period
        auto setup_track = [&](OscSlotTrack &D, const float *fa, const float *fb, int jEntry) {
            double x[4], e[4];
            osc_slot_exp_args(C, fa, fb, x);
            const double mine = exp2_d(part == 0 ? x[0] : part == 1 ? x[1] : part == 2 ? x[2] : x[3]);
            e[0] = q_take<3, kPart3>(q_take<2, kPart2>(q_take<1, kPart1>(mine, mine), mine), mine);     // slot 0's, in every slot
            e[1] = q_take<3, kPart0>(q_take<2, kPart3>(q_take<1, kPart2>(mine, mine), mine), mine);     // slot 1's
            e[2] = q_take<3, kPart1>(q_take<2, kPart0>(q_take<1, kPart3>(mine, mine), mine), mine);     // slot 2's
            e[3] = q_take<3, kPart2>(q_take<2, kPart1>(q_take<1, kPart0>(mine, mine), mine), mine);     // slot 3's
            osc_slot_from_exps<3>(D, C, fa, fb, jEntry, e);
        };
        if (nSteps > 0) {
            float fa[4], fb[4];
            frame_to(0, fa, 1);
            frame_to(1, fb, 1);
            setup_track(T, fa, fb, (int)j);
        }
        Tn = T;
        float2 *const ring = &sO[vq * kOStride];
        ScanState Z;
        scans_reset(Z);
        STAMP_DECL
        for (uint32_t step = 0; step < nSteps; step++) {
            STAMP_BEGIN
            // block i-2: the mix wave's {sig, thr} were written during step i-1; the tube wave reads the result from step i+1 on
            // (the scan in the mix wave instead -- the lighter role by instruction count -- measured 2.5 % slower)
            if (step >= 2 && (step - 2) * kOB < nTotal) {
                float &thr = x_at(kXThr, step - 2, slot);
                thr = throat_scan(Z, thr);
            }
            if (step * kOB < nTotal) {
                if (step * kOB + (kOB - 1) >= bnd) {        // (uniform) a lane enters the period that starts at sample bnd (:289)
                    const uint32_t ns = step * kOB + (uint32_t)slot;
                    float fa[4], fb[4];
                    frame_to(perN, fa, 1);
                    frame_to(perN + 1, fb, 1);
                    setup_track(Tn, fa, fb, (int)(ns >= bnd ? ns - bnd : ns + kOB - bnd));
                    perN++;
                    bnd += CP;
                }
                if (j >= CP) {      // this lane's sample starts a control period
                    j -= CP;
                    T = Tn;
                }
                const double db = __builtin_fma((double)j, T.glotDelta, T.glot0);
                double axd = db >= 60.0 ? 1.0 : T.axGeo;      // amplitude() with its clamps (:294-296)
                axd = db <= 0.0 ? 0.0 : axd;
                const float ah1 = amplitude_f(fma_f((float)j, T.aspDelta, T.aspBase));
                const double oinc = osc_increment(T.f0, C);       // (a multiple of 2^-30: the sums below are exact)
                // position after this lane's sample = P + the inclusive prefix sum of 2*inc over the step's slots: within
                // the block by row rotations, then the first block's total onto the second block's lanes
                double pre = oinc + oinc;
                pre += q_take<1, kPart1 | kPart2 | kPart3>(0.0, pre);
                pre += q_take<2, kPart2 | kPart3>(0.0, pre);
                double tot = pre;                                // slot 3's prefix = the block's advance
                tot = q_take<1, kPart0>(tot, pre);
                tot = q_take<2, kPart1>(tot, pre);
                tot = q_take<3, kPart2>(tot, pre);
                const unsigned long long tb = __builtin_bit_cast(unsigned long long, tot);
                const HalfPairU tl = across_halves((unsigned)tb), th = across_halves((unsigned)(tb >> 32));
                const double totLo = __builtin_bit_cast(double, ((unsigned long long)th.lo << 32) | tl.lo);
                const double totHi = __builtin_bit_cast(double, ((unsigned long long)th.hi << 32) | tl.hi);
                pre += upperHalf ? totLo : 0.0;
                const double end = P + pre;
                const double pos2 = osc_wrap(end), pos1 = osc_wrap(end - oinc);
                P = osc_wrap(P + (totLo + totHi));
                float wa, wb;
                osc_read(C, axd, pos1, pos2, sine, wa, wb);
                T.f0 *= T.f0Step;
                T.axGeo *= T.axStep;
                j += kOB;
                const uint32_t rs = (step * kOB + (uint32_t)slot) & (kORing - 1);
                ring[rs] = make_float2(wa, wb);
                if (rs < (uint32_t)kOMirror) ring[rs + kORing] = make_float2(wa, wb);
                sA[(step & 1u) * kWave + lane] = make_float2((float)axd, ah1);
            }
            STAMP_MID
            step_barrier();
            STAMP_END
        }
        STAMP_STORE(role)
    } else if (role == 1) {
#ifdef TRM_OCT_PRIO_MIX
        __builtin_amdgcn_s_setprio(TRM_OCT_PRIO_MIX);
#endif
        // ------------------------------------------------------------ mix: block i-1 at step i, lane = (voice, slot)
        auto fill_noise_half = [&](uint32_t nFirst, int half) {
            dma4(A.lp_noise + nFirst + lane, &sNoise[half * kNoiseHalf]);
        };
        // window taps of this lane's parity (m & 1 == slot & 1: steps start on multiples of 8), as (a, b) pairs
        const int o = slot & 1;
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
        // Converter coefficient rows, staged for the convert wave (see trm_quad.hip): block B's 32 rows are loaded when the
        // oscillator's time is within 4 samples of the block's first output, written to LDS one step later and visible
        // one step after that; sRowSync[1] = blocks staged, sRowSync[0] = the first block the convert wave has not copied yet.
        uint32_t rowBlk = 0;
        bool rowsInFlight = false;
        float4 rq[4];
        const uint32_t cvtOutputs = wave_max_u32(laneValid && nfr > 0 ? (uint32_t)((((uint64_t)ntubeLane + 2ull * (uint32_t)C.padSize) * 65536ull + inc - 1) / inc) : 0u);
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
            for (int r = 0; r < 2; r++) {
                if (rowBlk < cvtBlocks && src_position(rowBlk * kCvtCols, inc) <= step * kOB + 4u &&
                    rowBlk < lds_flag_consume(&sRowSync[0]) + kRowBufs) {
                    if (rowsInFlight) rows_to_lds();            // (a second block in the same step: its predecessor's loads are waited for here)
                    const uint32_t k = rowBlk * kCvtCols + ((uint32_t)lane >> 1);
                    const uint32_t off = (src_position(k, inc) + (kQLead - (kSrcWindow - 1))) & 3u;
                    const float *pc = A.src_rows + (size_t)src_phase(k, inc) * kSrcRowC - off + (lane & 1) * 16;
                    for (int q = 0; q < 4; q++) rq[q] = make_float4(pc[4 * q], pc[4 * q + 1], pc[4 * q + 2], pc[4 * q + 3]);
                    rowsInFlight = true;
                    rowBlk++;
                }
            }
            if (step >= 1 && (step - 1) * kOB < nTotal) {
                const uint32_t n0 = (step - 1) * kOB;
                if ((n0 & (kNoiseHalf - 1)) == 0 && n0 > 0) {
                    // entering a noise half: it was requested one half ago; refill the other half
                    dma_wait_all();
                    fill_noise_half(n0 + kNoiseHalf, ((n0 / kNoiseHalf) + 1) & 1);
                }
                const uint32_t m = n0 + (uint32_t)slot;
                // 26-sample window starting at the even index m - 24 - o (zeros before the first sample)
                const uint32_t s0 = (m + kORing - 24u - (uint32_t)o) & (kORing - 1);
                const float4 *wp = reinterpret_cast<const float4 *>(&ring[s0]);
                const float pulse = fir_window_dot(wp, cab);
                const float2 a = sA[((step - 1) & 1u) * kWave + lane];
                const Excitation E = mix_tail(C, a.x, a.y, pulse, sNoise[m & (kNoiseRing - 1)]);
                x_at(kXGin, step - 1, slot) = E.gin;
                x_at(kXSig, step - 1, slot) = E.sig;
                x_at(kXThr, step - 1, slot) = E.thr;       // (raw: the oscillator wave turns it into the throat output one step on)
            }
            STAMP_MID
            step_barrier();
            STAMP_END
        }
        STAMP_STORE(role)
        dma_wait_all();
    } else if (role == 2 || role == 3) {
#ifdef TRM_OCT_PRIO_COEF
        if (role == 2) __builtin_amdgcn_s_setprio(TRM_OCT_PRIO_COEF);
#endif
#ifdef TRM_OCT_PRIO_COEF2
        if (role == 3) __builtin_amdgcn_s_setprio(TRM_OCT_PRIO_COEF2);
#endif
        // ------------------------------------------------------------ coef: block i-2 at step i, lane = (voice, slot).
        // Role 2 turns radii and velum into the junctions' transmission factors (stateless in time), role 3 the frication
        // tracks into taps and band-pass coefficients, runs the band-pass over the mix wave's noise signal (written during
        // step i-1) and hands the tube the INJECTIONS tap x band-pass output.
        const bool area = role == 2;
        ScanState Z;
        scans_reset(Z);
        CoefTrack T;
        uint32_t per = 0, j = (uint32_t)slot;      // the period between frames per and per + 1
        if (nSteps > 0) {
            float fa[16], fb[16];
            frame_to(0, fa, 4);
            frame_to(1, fb, 4);
            coef_track_setup(T, C, fa, fb);
        }
        // the stager (area wave): in the step in which the oscillator wave enters the period between frames p and p + 1
        // (it is two blocks ahead of this wave), the transfer of frame p + 2 is retired and frame p + 3 sent for
        uint32_t stP = 1, stBnd = CP;
        STAMP_DECL
        for (uint32_t step = 0; step < nSteps; step++) {
            STAMP_BEGIN
            if (area && step * kOB < nTotal && step * kOB + (kOB - 1) >= stBnd) {
                dma_wait_all();
                stage_frame(stP + 3);
                stP++;
                stBnd += CP;
            }
            if (step >= 2 && (step - 2) * kOB < nTotal) {
                const uint32_t blk = step - 2;
                if (j >= CP) {
                    j -= CP;
                    per++;
                    float fa[16], fb[16];
                    frame_to(per, fa, 4);
                    frame_to(per + 1, fb, 4);
                    coef_track_setup(T, C, fa, fb);
                }
                Coefs K;
                // [buf][slot][part p][voice]{k.x, k.y | in.x, in.y}: the area wave writes the k halves (and the mouth
                // end's 1 + C8), the frication wave the injections
                float4 *row = &sK[((blk % kKDepth) * kOB + slot) * kOKRow];
                if (area) {
                    coef_sample_area(K, T, C, (int)j);
                    float kk[8][2];
                    pack_oct_k(K, C, kk);
                    for (int p = 0; p < 6; p++) reinterpret_cast<float2 *>(&row[oct_k_index(p, vq)])[0] = make_float2(kk[p][0], kk[p][1]);
                    reinterpret_cast<float *>(&row[oct_k_index(4, vq)])[3] = K.onePlusK8;
                    x_at(kXAlpha, blk, slot) = K.alphaLR;     // the three-way junction's alpha-left = alpha-right
                } else {
                    coef_sample_fric(K, T, C, (int)j);
                    float tp[5][2];
                    pack_oct_tap(K, tp);
                    SharedRecord H;
                    pack_shared_bp(K, H);
                    const float sig = x_at(kXSig, blk, slot);
                    const float f = bandpass_scan(Z, sig, make_float4(H.bpA2, H.bpB2, H.bpG2, 0.0f));
                    for (int p = 0; p < 4; p++) reinterpret_cast<float2 *>(&row[oct_k_index(p, vq)])[1] = make_float2(tp[p][0] * f, tp[p][1] * f);
                    reinterpret_cast<float *>(&row[oct_k_index(4, vq)])[2] = tp[4][0] * f;
                }
                j += kOB;
            }
            STAMP_MID
            step_barrier();
            STAMP_END
        }
        STAMP_STORE(role)
        if (area) dma_wait_all();   // nothing may still be writing LDS when the wave ends
    } else if (role == 4) {
        // ------------------------------------------------------------ tube: block i-4 at step i, lane = (voice, part)
        // the only serial role: its instructions go first on the SIMD it shares with the other workgroup's waves
        // (2.89 -> 2.74 ms with the first role assignment, 3.21 -> 2.58 with this one)
#ifndef TRM_OCT_TUBE_PRIO
#define TRM_OCT_TUBE_PRIO 3
#endif
        __builtin_amdgcn_s_setprio(TRM_OCT_TUBE_PRIO);
        const int pT = lane & 7;
        OctLane<float> OL;
        OL.p0 = pT == 0; OL.p1 = pT == 1; OL.p5 = pT == 5; OL.end = pT == 4 || pT == 7;
        OL.cf = pT == 4 ? C.mCoeff : pT == 7 ? C.nCoeff : 0.0f;
        // (vector-register copies: an instruction with a scalar operand does not co-issue with the SIMD's other waves)
        float dV = C.damping, tgV = C.throatGain;
        asm volatile("" : "+v"(dV), "+v"(tgV));
        OctState<float> S;
        oct_reset(S);
        float4 *const ring = reinterpret_cast<float4 *>(&sY[vq * kYStride]);
        float *const tubeOut = A.tube_out ? A.tube_out + A.tube_offset[v] : nullptr;
        // one sample's inputs: this part's record {k | injections} and the ONE per-voice value this part uses (part 0: the
        // glottal input, part 1: the three-way junction's alpha, part 4: the throat output)
        struct In { float4 r; float xs; };
        const int xArr = pT == 0 ? kXGin : pT == 1 ? kXAlpha : kXThr;
        const int kIdx = oct_k_index(pT, vq);
        auto load_in = [&](uint32_t blk, int s) {
            In r;
            r.r = sK[((blk % kKDepth) * kOB + s) * kOKRow + kIdx];
            r.xs = x_at(xArr, blk, s);
            return r;
        };
        auto step_one = [&](const In &r) {
            return tube_oct_core<float>(S, dV, tgV, OL, r.xs, r.xs, r.xs, v2f_t{r.r.x, r.r.y}, v2f_t{r.r.z, r.r.w});
        };
        // A block's first sample's inputs are fetched behind the last sample of the block before it (during step i-1),
        // the others land behind the first samples' arithmetic: no LDS latency is exposed.
        In head;
        head.r = make_float4(0.f, 0.f, 0.f, 0.f);
        head.xs = 0.f;
        STAMP_DECL
        for (uint32_t step = 0; step < nSteps; step++) {
            STAMP_BEGIN
            if (step >= 4 && (step - 4) * kOB < nTotal) {
                const uint32_t blk = step - 4, n0 = blk * kOB;
                float y[kOB];
                // (samples past nTotal in the last block step on stale inputs; their output is forced to 0)
                {
                    const In i1 = load_in(blk, 1), i2 = load_in(blk, 2), i3 = load_in(blk, 3);
                    y[0] = step_one(head);
                    const In i4 = load_in(blk, 4);
                    y[1] = step_one(i1);
                    const In i5 = load_in(blk, 5);
                    y[2] = step_one(i2);
                    const In i6 = load_in(blk, 6);
                    y[3] = step_one(i3);
                    const In i7 = load_in(blk, 7);
                    y[4] = step_one(i4);
                    y[5] = step_one(i5);
                    y[6] = step_one(i6);
                    head = load_in(blk + 1, 0);
                    y[7] = step_one(i7);
                }
                if (n0 + kOB > ntubeMin) {     // (uniform) zero flush / voices shorter than the group's longest
                    asm volatile("");          // (a real branch: if-converted, the selects run on every sample of every step)
#pragma unroll
                    for (int s = 0; s < kOB; s++) y[s] = n0 + s < ntubeLane ? y[s] : 0.0f;
                }
                if (pT == 4) {
                    // tube sample n sits at ring slot (n + kQLead) & 127: a block of four is one aligned 16-byte store
#pragma unroll
                    for (int h = 0; h < 2; h++) {
                        const uint32_t nh = n0 + 4u * h;
                        const uint32_t slot4 = ((nh + kQLead) & (kYRing - 1)) >> 2;
                        const float4 yy = make_float4(y[4 * h], y[4 * h + 1], y[4 * h + 2], y[4 * h + 3]);
                        ring[slot4] = yy;
                        if (slot4 < (uint32_t)(kYMirror / 4)) ring[slot4 + kYRing / 4] = yy;
                        if (tubeOut && laneValid) {
                            // (rows of tube_out are 16-byte aligned: the host pads their pitch to 4 floats)
                            const uint32_t lim = ntubeLane + 2u * (uint32_t)C.padSize;
                            if (nh + 4u <= lim) *reinterpret_cast<float4 *>(tubeOut + nh) = yy;
                            else
                                for (int s = 0; s < 4; s++)
                                    if (nh + s < lim) tubeOut[nh + s] = y[4 * h + s];
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
        // (tube 3 > convert 2 > oscillator 1 > the rest: 2.41 -> 2.375 ms at the bench batch, profiles/ab_r03.txt)
#ifndef TRM_OCT_PRIO_CVT
#define TRM_OCT_PRIO_CVT 2
#endif
        __builtin_amdgcn_s_setprio(TRM_OCT_PRIO_CVT);
        // ------------------------------------------------------------ convert (lane = output time), 8 voices
        uint32_t noutLane = 0;
        if (nfr > 0) {
            uint64_t total = (uint64_t)ntubeLane + 2ull * (uint32_t)C.padSize;
            noutLane = (uint32_t)((total * 65536ull + inc - 1) / inc);
        }
        if (!laneValid) noutLane = 0;
        const uintptr_t myOut = reinterpret_cast<uintptr_t>(A.out + A.out_offset[v]);
        const uint32_t noutMax = wave_max_u32(noutLane);
        const uint32_t nBlocks = C.upsample ? (noutMax + kCvtCols - 1) / kCvtCols : 0;
        const int col = lane & (kCvtCols - 1);
        if (part == 0 && !upperHalf) sInfo[vq] = make_uint4(noutLane, (uint32_t)myOut, (uint32_t)(myOut >> 32), 0u);
        // running max |y| per (row, lane): row r = voices 2r (lanes 0-31) and 2r+1 (lanes 32-63)
        for (int r = 0; r < 4; r++) sMx[r * kWave + lane] = 0.0f;

        // output k reads tube samples e-25 .. e, e = src_position(k): ring slots e + kRingShift .. + 25
        constexpr uint32_t kRingShift = kQLead - (kSrcWindow - 1);
        v2f cc[16];
        uint32_t blk = 0, pr = 0;       // next work item: row pair `pr` (0..1) of block `blk`: voices 4*pr .. 4*pr+3
        uint32_t winBase = 0, kLane = 0, needLast = 0, needNext = 0;
        auto begin_block = [&]() {
            kLane = blk * kCvtCols + col;
            winBase = (src_position(kLane, inc) + kRingShift) & (kYRing - 1) & ~3u;
            needLast = src_position(blk * kCvtCols + (kCvtCols - 1), inc);
            needLast = needLast < nTotal - 1 ? needLast : nTotal - 1;
            needNext = src_position((blk + 1) * kCvtCols + (kCvtCols - 1), inc);   // (past the end: never "behind")
            const float4 *row = reinterpret_cast<const float4 *>(&sRows[(blk % kRowBufs) * (kCvtCols * kRowPitch) + col * kRowPitch]);
            for (int q = 0; q < 8; q++) {
                const float4 x = row[q];
                cc[2 * q] = v2f{x.x, x.y};
                cc[2 * q + 1] = v2f{x.z, x.w};
            }
        };
        auto begin_block_from_global = [&]() {
            kLane = blk * kCvtCols + col;
            winBase = (src_position(kLane, inc) + kRingShift) & (kYRing - 1) & ~3u;
            needLast = nTotal - 1;
            const uint32_t off = (src_position(kLane, inc) + kRingShift) & 3u;
            const float *pc = A.src_rows + (size_t)src_phase(kLane, inc) * kSrcRowC - off;
            for (int q = 0; q < 16; q++) cc[q] = v2f{pc[2 * q], pc[2 * q + 1]};
        };
        bool needBegin = nBlocks > 0;
        typedef __attribute__((address_space(1))) float *GlobalFloatPtr;
        typedef __attribute__((address_space(3))) float *LdsFloatPtr;
        // metering (16.16 row pairs per step): a step's kOB tube samples turn into kOB * 2^16/inc outputs per voice =
        // that / 32 blocks of 2 row pairs
        const uint32_t earn = (uint32_t)(((uint64_t)kOB << 32) / inc / 16) + 2048;
        // the cap must leave room to catch up after waiting for a block (see trm_quad.hip)
        const uint32_t capPairs = (earn + 0x18000u) >> 16;                  // floor(earn + 1.5)
        const uint32_t creditCap = (capPairs > 2u ? capPairs : 2u) << 16;
        uint32_t credit = 0;
        auto do_pair = [&]() {
            const int la = 4 * (int)pr, lb = la + 2;
            const int ha = upperHalf ? 1 : 0;
            const float4 *wa = reinterpret_cast<const float4 *>(&sY[(la + ha) * kYStride + winBase]);
            const float4 *wb = reinterpret_cast<const float4 *>(&sY[(lb + ha) * kYStride + winBase]);
            const uint4 ia = sInfo[la + ha], ib = sInfo[lb + ha];
            // (the second row's window is read while the first row's chains run: one row's registers are live at a time)
            const float ya = cvt_window_dot(wa, cc), yb = cvt_window_dot(wb, cc);
            const bool okA = kLane < ia.x, okB = kLane < ib.x;
            if (okA) reinterpret_cast<GlobalFloatPtr>(((uintptr_t)ia.z << 32) | ia.y)[kLane] = ya;
            if (okB) reinterpret_cast<GlobalFloatPtr>(((uintptr_t)ib.z << 32) | ib.y)[kLane] = yb;
            __builtin_amdgcn_ds_fmaxf((LdsFloatPtr)&sMx[(2 * pr) * kWave + lane], okA ? fabsf(ya) : 0.0f, 0, 0, false);
            __builtin_amdgcn_ds_fmaxf((LdsFloatPtr)&sMx[(2 * pr + 1) * kWave + lane], okB ? fabsf(yb) : 0.0f, 0, 0, false);
            if (++pr == 2) {
                pr = 0;
                blk++;
                needBegin = blk < nBlocks;
            }
        };
        STAMP_DECL
        for (uint32_t step = 0; step < nSteps; step++) {
            STAMP_BEGIN
            // visible after the previous barrier: tube samples n < (step-4)*kOB
            const uint32_t ready = step >= 4 ? (step - 4) * kOB : 0;
            credit += earn;
            if (credit > creditCap) credit = creditCap;     // a ready block is spread over the next steps, not done in a burst
            auto try_begin = [&]() {
                if (blk < lds_flag_consume(&sRowSync[1])) {
                    begin_block();
                    needBegin = false;
                    lds_flag_publish(&sRowSync[0], blk + 1, lane == 0);    // this block's rows are in registers now
                }
            };
            if (needBegin) try_begin();
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
        float myMax = 0.0f;     // collected by lanes 0..7: voice `lane` of the workgroup
#pragma unroll
        for (int r = 0; r < 4; r++) {
            float m = sMx[r * kWave + lane];
            for (int off = 16; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, kWave));
            const float lowHalf = __shfl(m, 0, kWave), highHalf = __shfl(m, 32, kWave);
            if (lane == 2 * r) myMax = lowHalf;
            if (lane == 2 * r + 1) myMax = highHalf;
        }
        const uint32_t ov = blockIdx.x * kOV + (uint32_t)lane;
        if (lane < kOV && ov < A.nvoices && C.upsample) {
            const uint32_t nf = min(A.nframes[ov], A.max_nframes);
            uint32_t nov = 0;
            if (nf > 0) nov = (uint32_t)((((uint64_t)(nf - 1) * CP + 2ull * (uint32_t)C.padSize) * 65536ull + inc - 1) / inc);
            A.number_samples[ov] = nov;
            A.max_sample[ov] = myMax;
        }
        return;
    }
}

hipError_t launch_tube_oct(const Const &c, const TubeArgs &a, hipStream_t stream)
{
    if (a.nvoices == 0) return hipSuccess;
    if (a.stream_state || c.controlPeriod < 2 * kOB) return hipErrorInvalidValue;     // (the caller picks trm_quad.hip's kernel for these)
    static DynamicLdsAllowance lds;
    hipError_t e = lds.ensure(reinterpret_cast<const void *>(trm_tube_kernel_o), (int)OctLds::kBytes);
    if (e != hipSuccess) return e;
    const uint32_t grid = (a.nvoices + kOV - 1) / kOV;
    hipLaunchKernelGGL(trm_tube_kernel_o, dim3(grid), dim3(kWave * kORoles), OctLds::kBytes, stream, c, a);
    return hipGetLastError();
}

int tube_oct_kernel_blocks_per_cu()
{
    int n = 0;
    return hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, trm_tube_kernel_o, kWave * kORoles, OctLds::kBytes) == hipSuccess ? n : -1;
}

}  // namespace trm
