// trm_kernels.hip -- CDNA4 (gfx950) kernels of the Tube Resonance Model.
//
//   trm_noise_kernel      the voice-independent noise sequence (TRMUtility.m:71-85 + TRMFilters.m:81-86),
//                         fp64 serial recurrence, one lane, run once per batch object and cached
//   trm_tube_kernel       -[TRMTubeModel synthesize] (TRMTubeModel.m:272-361): one tube per lane, 64 voices
//                         per workgroup, 4 waves per workgroup running the sample loop as a pipeline
//                         (excite | coef | tube | convert) with hand-offs through LDS; state in VGPRs;
//                         wave-uniform control in SGPRs; converter coefficients + noise prefetched into
//                         LDS rings by LDS-DMA one half ahead; output staged through LDS, written as rows
//   trm_int16_kernel      output normalisation (TRMTubeModel.m:370-389, 420-484)
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "trm_kernels.h"
#include "trm_lane.h"

#ifndef TRM_ABL
#define TRM_ABL 0    // diagnostic ablations of the convert stage (tools/stage_profile.py); 0 in the product
#endif

namespace trm {

constexpr int kWave = 64;
constexpr int kRoles = 4;            // waves per workgroup: excite, coef, tube, convert
constexpr int kTB = 2;               // tube samples per pipeline step (one barrier per step)
constexpr int kTile = 16;            // outputs per staged half-tile; the staging ring holds two halves
constexpr int kTileStride = 2 * kTile + 4;   // 16-byte aligned rows for b128 row reads
constexpr int kRowSlots = 64;        // converter coefficient ring: one slot per output sample
constexpr int kRowHalf = 32;         //   refilled by halves, one half ahead
constexpr int kSlotFloats = 32;      //   slot = left-wing row (16 floats) + right-wing row (16 floats)
constexpr int kNoiseRing = 128;      // noise ring: one float per tube sample, refilled by halves of 64
constexpr int kNoiseHalf = 64;

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

// ---------------------------------------------------------------- LDS-DMA helpers
// global_load_lds_*: asynchronous global -> LDS copy, no VGPR destination.  The LDS address is a
// wave-uniform base (M0) + lane * size; the global source address is per lane.  Completion is
// tracked by vmcnt; the compiler does not know these writes, so readers wait explicitly.
typedef __attribute__((address_space(1))) const void *GlobalPtr;
typedef __attribute__((address_space(3))) void *LdsPtr;
// 16-byte vector with 4-byte alignment: PCM rows start at arbitrary sample offsets
typedef float float4u __attribute__((ext_vector_type(4), aligned(4)));

__device__ __forceinline__ void dma16(const float *src, float *ldsBaseUniform)
{
    __builtin_amdgcn_global_load_lds((GlobalPtr)src, (LdsPtr)ldsBaseUniform, 16, 0, 0);
}
__device__ __forceinline__ void dma4(const float *src, float *ldsBaseUniform)
{
    __builtin_amdgcn_global_load_lds((GlobalPtr)src, (LdsPtr)ldsBaseUniform, 4, 0, 0);
}
__device__ __forceinline__ void dma_wait_all() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v)
{
    for (int off = 32; off > 0; off >>= 1) {
        uint32_t o = __shfl_xor(v, off, kWave);
        v = o > v ? o : v;
    }
    return __builtin_amdgcn_readfirstlane(v);
}

// Diagnostic build only (-DTRM_STAMP, tools/stage_profile.py): per-role cycles spent working vs
// waiting at the step barrier.  In the product build these macros expand to nothing.
#ifdef TRM_STAMP
#define STAMP_DECL unsigned long long st_work = 0, st_wait = 0, st_t0 = 0, st_t1 = 0;
#define STAMP_BEGIN st_t0 = __builtin_readcyclecounter();
#define STAMP_MID st_t1 = __builtin_readcyclecounter(); st_work += st_t1 - st_t0;
#define STAMP_END st_wait += __builtin_readcyclecounter() - st_t1;
#define STAMP_STORE(role_)                                                          \
    if (lane == 0 && A.stamps) {                                                    \
        A.stamps[(blockIdx.x * kRoles + (role_)) * 2] = st_work;                    \
        A.stamps[(blockIdx.x * kRoles + (role_)) * 2 + 1] = st_wait;                \
    }
#else
#define STAMP_DECL
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

// One workgroup = 64 voices (one per lane) x 4 waves (one per pipeline stage).  At step i the
// excite and coef waves produce block i (kTB tube samples) into LDS, the tube wave consumes block
// i-1 and produces tube-rate samples, the convert wave consumes block i-2 and writes PCM.  One
// barrier per step; every hand-off buffer is double-buffered.
__global__ __launch_bounds__(kWave *kRoles) void trm_tube_kernel(const Const C, const TubeArgs A)
{
    __shared__ __attribute__((aligned(16))) float4 sX[2 * kTB * kWave];          // excitation per sample
    __shared__ __attribute__((aligned(16))) float4 sK[2 * kTB * 6 * kWave];      // coefficients per sample
    __shared__ float sY[2 * kTB * kWave];                                        // tube-rate samples
    __shared__ float sStage[kWave * kTileStride];                                // convert: output tile
    __shared__ __attribute__((aligned(16))) float sRows[kRowSlots * kSlotFloats]; // convert: coefficient ring
    __shared__ float *sOutPtr[kWave];                                            // convert: row destinations
    __shared__ uint32_t sOutLen[kWave];                                          // convert: row lengths
    __shared__ float sNoise[kNoiseRing];                                         // excite: noise ring
    __shared__ float sFir[32];                                                   // excite: FIR taps
    __shared__ float sSine[kTableLen];                                           // excite: sine table

    const int lane = threadIdx.x & (kWave - 1);
    const int role = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t vRaw = blockIdx.x * kWave + lane;
    const bool laneValid = vRaw < A.nvoices;
    const uint32_t v = laneValid ? vRaw : A.nvoices - 1;

    const uint32_t nfr = A.nframes[v];
    const uint32_t nfrMax = wave_max_u32(nfr);          // same 64 voices in every wave of the group
    const uint32_t CP = (uint32_t)C.controlPeriod;
    const uint32_t ntubeMax = nfrMax > 0 ? (nfrMax - 1) * CP : 0;
    // (nfrMax-1) control periods, then the converter's 2*pad zero flush (TRMRingBuffer.m:85-93).
    // Lanes whose utterance is shorter than the group's longest keep stepping on their last frame;
    // the convert stage forces their converter input to 0 and masks their stores.
    const uint32_t nTotal = nfrMax > 0 ? ntubeMax + 2u * (uint32_t)C.padSize : 0;
    const uint32_t nSteps = nTotal > 0 ? (nTotal + kTB - 1) / kTB + 2 : 0;
    // a voice without frames (a silent no-op, TRMTubeModel.m:274-277) reads row 0 of the buffer
    const float *frames = A.frames + (nfr > 0 ? A.frame_offset[v] * 16 : 0);

    if (role == 0) {
        // ------------------------------------------------------------ excite
        if (C.waveform != 0)
            for (int i = lane; i < kTableLen; i += kWave) sSine[i] = A.sine[i];
        auto sine = [&](int i) { return sSine[i]; };
        auto fill_noise_half = [&](uint32_t nFirst, int half) {
            dma4(A.lp_noise + nFirst + lane, &sNoise[half * kNoiseHalf]);
        };
        ExciteState S;
        ExciteTrack T;
        excite_reset(S);
        // The 25 distinct FIR taps live in VGPRs of this wave (uniform values): as SGPRs they would
        // exceed the scalar file together with the other constants and be spilled to VGPR lanes.
        for (int i = lane; i < kFirUnique; i += kWave) sFir[i] = C.fir[i];
        float firv[kFirUnique];
        for (int i = 0; i < kFirUnique; i++) firv[i] = sFir[i];
        float cur[4], prev[4];
        if (nSteps > 0) {
            load_frame(frames, 0, cur, 1);
            fill_noise_half(0, 0);
            fill_noise_half(kNoiseHalf, 1);
            dma_wait_all();
        }
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
                        for (int q = 0; q < 4; q++) prev[q] = cur[q];
                        load_frame(frames, f < nfr ? f : (nfr > 0 ? nfr - 1 : 0), cur, 1);
                        excite_track_setup(T, C, prev, cur);
                    }
                    if ((n & (kNoiseHalf - 1)) == 0 && n > 0) {
                        // entering a noise half: it was requested one half ago; refill the other half
                        dma_wait_all();
                        fill_noise_half(n + kNoiseHalf, ((n / kNoiseHalf) + 1) & 1);
                    }
                    Excitation E = excite_sample(S, T, C, firv, (int)j, sNoise[n & (kNoiseRing - 1)], sine);
                    j++;
                    sX[(buf * kTB + u) * kWave + lane] = make_float4(E.gin, E.sig, E.thr, 0.0f);
                }
            }
            STAMP_MID
            __syncthreads();
            STAMP_END
        }
        STAMP_STORE(role)
        dma_wait_all();   // nothing may still be writing LDS when the wave ends
    } else if (role == 1) {
        // ------------------------------------------------------------ coef
        CoefTrack T;
        float cur[16], prev[16];
        if (nSteps > 0) load_frame(frames, 0, cur, 4);
        uint32_t j = CP, f = 0;
        STAMP_DECL
        for (uint32_t step = 0; step < nSteps; step++) {
            STAMP_BEGIN
            const int buf = step & 1;
            for (int u = 0; u < kTB; u++) {
                const uint32_t n = step * kTB + u;
                if (n < nTotal) {
                    if (j == CP) {
                        j = 0;
                        f++;
                        for (int q = 0; q < 16; q++) prev[q] = cur[q];
                        load_frame(frames, f < nfr ? f : (nfr > 0 ? nfr - 1 : 0), cur, 4);
                        coef_track_setup(T, C, prev, cur);
                    }
                    Coefs K = coef_sample(T, C, (int)j);
                    j++;
                    float4 *dst = &sK[((buf * kTB + u) * 6) * kWave + lane];
                    dst[0 * kWave] = make_float4(K.k[0], K.k[1], K.k[2], K.k[3]);
                    dst[1 * kWave] = make_float4(K.k[4], K.k[5], K.k[6], K.k[7]);
                    dst[2 * kWave] = make_float4(K.onePlusK8, K.alphaLR, K.alphaU, K.nk1);
                    dst[3 * kWave] = make_float4(K.tap[0], K.tap[1], K.tap[2], K.tap[3]);
                    dst[4 * kWave] = make_float4(K.tap[4], K.tap[5], K.tap[6], K.tap[7]);
                    dst[5 * kWave] = make_float4(K.bpAlpha, K.bpBeta, K.bpGamma, 0.0f);
                }
            }
            STAMP_MID
            __syncthreads();
            STAMP_END
        }
        STAMP_STORE(role)
    } else if (role == 2) {
        // ------------------------------------------------------------ tube
        TubeState S;
        tube_reset(S);
        STAMP_DECL
        for (uint32_t step = 0; step < nSteps; step++) {
            STAMP_BEGIN
            if (step >= 1) {
                const uint32_t blk = step - 1;
                const int buf = blk & 1;
                for (int u = 0; u < kTB; u++) {
                    const uint32_t n = blk * kTB + u;
                    if (n < nTotal) {
                        const float4 x = sX[(buf * kTB + u) * kWave + lane];
                        const float4 *src = &sK[((buf * kTB + u) * 6) * kWave + lane];
                        const float4 k0 = src[0 * kWave], k1 = src[1 * kWave], k2 = src[2 * kWave];
                        const float4 t0 = src[3 * kWave], t1 = src[4 * kWave], bp = src[5 * kWave];
                        Excitation E;
                        E.gin = x.x; E.sig = x.y; E.thr = x.z;
                        Coefs K;
                        K.k[0] = k0.x; K.k[1] = k0.y; K.k[2] = k0.z; K.k[3] = k0.w;
                        K.k[4] = k1.x; K.k[5] = k1.y; K.k[6] = k1.z; K.k[7] = k1.w;
                        K.onePlusK8 = k2.x; K.alphaLR = k2.y; K.alphaU = k2.z; K.nk1 = k2.w;
                        K.tap[0] = t0.x; K.tap[1] = t0.y; K.tap[2] = t0.z; K.tap[3] = t0.w;
                        K.tap[4] = t1.x; K.tap[5] = t1.y; K.tap[6] = t1.z; K.tap[7] = t1.w;
                        K.bpAlpha = bp.x; K.bpBeta = bp.y; K.bpGamma = bp.z; K.pad_ = 0.0f;
                        sY[(buf * kTB + u) * kWave + lane] = tube_sample(S, C, E, K);
                    }
                }
            }
            STAMP_MID
            __syncthreads();
            STAMP_END
        }
        STAMP_STORE(role)
    } else {
        // ------------------------------------------------------------ convert
        float *const outBase = A.out + A.out_offset[v];
        const uint32_t inc = C.timeRegisterIncrement;
        const uint32_t ntubeLane = nfr > 0 ? (nfr - 1) * CP : 0;
        uint32_t noutLane = 0;
        if (nfr > 0) {
            uint64_t total = (uint64_t)ntubeLane + 2ull * (uint32_t)C.padSize;
            noutLane = (uint32_t)((total * 65536ull + inc - 1) / inc);
        }
        if (!laneValid) noutLane = 0;
        sOutPtr[lane] = outBase;         // read back row-wise by the tile flush (this wave only)
        sOutLen[lane] = noutLane;
        SrcState<kTB> S;
        src_reset(S);
        // Converter coefficients for output k live in slot k & 63.  Output k's phase is (k*inc) mod 2^16
        // (TRMSampleRateConverter.m:221-232), so rows can be fetched ahead by output index alone.
        // One DMA instruction fills 8 slots: lane -> slot (lane>>3), 16-byte part (lane&7) = {L q0..3, R q0..3}.
        auto fill_rows_half = [&](uint32_t kFirst, int half) {
            for (int jj = 0; jj < 4; jj++) {
                uint32_t k = kFirst + (uint32_t)(jj * 8 + (lane >> 3));
                uint32_t ph = (k * inc) & 0xFFFFu;
                uint32_t row = (lane & 4) ? 0xFFFFu - ph : ph;
                dma16(A.src_rows + (size_t)row * kSrcRow + (lane & 3) * 4,
                      &sRows[(half * kRowHalf + jj * 8) * kSlotFloats]);
            }
        };
        // Two coefficient register sets used alternately inside a step; set A always holds the rows of the
        // step's first output, fetched at the end of the previous step so the LDS latency hides behind the
        // barrier.  Inside the step the rows of output o+1 are requested before the 26 FMAs of output o.
        float cA[32], cB[32];
        auto load_rows = [&](float *c, uint32_t k) {
            const float4 *rp = reinterpret_cast<const float4 *>(&sRows[(k & (kRowSlots - 1)) * kSlotFloats]);
            for (int q = 0; q < 8; q++) {
                float4 a = rp[q];
                c[4 * q] = a.x; c[4 * q + 1] = a.y; c[4 * q + 2] = a.z; c[4 * q + 3] = a.w;
            }
        };
        if (nSteps > 0) {
            fill_rows_half(0, 0);
            fill_rows_half(kRowHalf, 1);
            dma_wait_all();
            load_rows(cA, 0);
        }
        uint32_t e = 0;           // converter read position, in pushed samples (uniform)
        uint32_t t = 0;           // 16.16 time register, N part cleared (uniform)
        uint32_t kout = 0;        // outputs emitted (uniform)
        uint32_t kflushed = 0;    // outputs written to HBM (uniform, multiple of kTile)
        uint32_t rowsHalfDone = 0;   // ring halves already re-requested (uniform)

        // outputs belonging to tube sample n: those k with floor(k*inc / 2^16) == n (uniform, scalar only)
        auto count_outputs = [&](uint32_t n) {
            uint32_t c = 0;
            while (e <= n) {
                t += inc;
                e += t >> 16;
                t &= 0xFFFFu;
                c++;
            }
            return c;
        };
        auto emit_one = [&](const float *cur, float *nxt, bool second) {
#if TRM_ABL != 2
            load_rows(nxt, kout + 1);
#endif
#if TRM_ABL == 3
            float y = cur[0] + cur[16] + S.src[second ? 1 : 0];
#else
            float y = second ? src_emit_up<kTB, 1>(S, cur, cur + 16) : src_emit_up<kTB, 0>(S, cur, cur + 16);
#endif
            float a = fabsf(y);
            S.maxAbs = (kout < noutLane && a > S.maxAbs) ? a : S.maxAbs;
            sStage[lane * kTileStride + (kout & (2 * kTile - 1))] = y;
            kout++;
        };
        // one half (kTile outputs) of the staging ring -> HBM: 16 rows per pass, lane -> row (lane>>2),
        // 16-byte piece (lane&3); a row is 64 contiguous bytes
        auto flush_half = [&]() {
#if TRM_ABL == 1
            kflushed += kTile;
            return;
#endif
            const uint32_t c4 = (uint32_t)(lane & 3) * 4u;
            const uint32_t col = (kflushed & (2 * kTile - 1)) + c4;
            const uint32_t k = kflushed + c4;
#pragma unroll
            for (int it = 0; it < kWave / 16; it++) {
                const int row = it * 16 + (lane >> 2);
                float *dst = sOutPtr[row] + k;
                const uint32_t lim = sOutLen[row];
                const float4 val = *reinterpret_cast<const float4 *>(&sStage[row * kTileStride + col]);
                if (k + 3 < lim && (reinterpret_cast<uintptr_t>(dst) & 15) == 0) {
                    *reinterpret_cast<float4 *>(dst) = val;
                } else {
                    if (k < lim) dst[0] = val.x;
                    if (k + 1 < lim) dst[1] = val.y;
                    if (k + 2 < lim) dst[2] = val.z;
                    if (k + 3 < lim) dst[3] = val.w;
                }
            }
            kflushed += kTile;
        };

        STAMP_DECL
        for (uint32_t step = 0; step < nSteps; step++) {
            STAMP_BEGIN
            if (step >= 2) {
                const uint32_t blk = step - 2;
                const int buf = blk & 1;
                float sv[kTB];
                for (int u = 0; u < kTB; u++) {
                    const uint32_t n = blk * kTB + u;
                    float s = sY[(buf * kTB + u) * kWave + lane];
                    sv[u] = (n < ntubeLane && n < nTotal) ? s : 0.0f;
                }
                src_push_block<kTB>(S, sv);
                // about to read coefficient slots of the next ring half: its DMA (issued most of a half
                // ago) must have landed
                if (((kout + 8) / kRowHalf) > rowsHalfDone) dma_wait_all();
                const uint32_t n0 = blk * kTB;
                const uint32_t cnt0 = n0 < nTotal ? count_outputs(n0) : 0;
                const uint32_t cntAll = cnt0 + (n0 + 1 < nTotal ? count_outputs(n0 + 1) : 0);
                // straight-line emission, register sets alternate A, B, A, ... from the step's first output
                for (uint32_t o = 0; o < cntAll; o += 2) {
                    emit_one(cA, cB, o >= cnt0);
                    if (o + 1 < cntAll) emit_one(cB, cA, o + 1 >= cnt0);
                    else load_rows(cA, kout);      // odd count: the next step starts from set A again
                }
                // single-site housekeeping: coefficient ring refill, staged tile flush
                if ((kout / kRowHalf) > rowsHalfDone) {
                    // a ring half has been fully consumed: wait for the DMA issued one half ago, refill it
                    dma_wait_all();
                    rowsHalfDone++;
                    fill_rows_half((rowsHalfDone + 1) * kRowHalf, (rowsHalfDone + 1) & 1);
                }
                if (kout - kflushed >= kTile) flush_half();
            }
            STAMP_MID
            __syncthreads();
            STAMP_END
        }
        STAMP_STORE(role)
        while (kflushed < kout) flush_half();     // tail (rows are masked by their own lengths)
        dma_wait_all();
        if (laneValid) {
            A.number_samples[vRaw] = noutLane;
            A.max_sample[vRaw] = S.maxAbs;
        }
    }
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

// ---------------------------------------------------------------- launchers (host)
hipError_t launch_noise(float *lp, uint32_t from, uint32_t to, double *state, hipStream_t stream)
{
    hipLaunchKernelGGL(trm_noise_kernel, dim3(1), dim3(1), 0, stream, lp, from, to, state);
    return hipGetLastError();
}

hipError_t launch_tube(const Const &c, const TubeArgs &a, hipStream_t stream)
{
    if (a.nvoices == 0) return hipSuccess;
    uint32_t grid = (a.nvoices + kWave - 1) / kWave;
    hipLaunchKernelGGL(trm_tube_kernel, dim3(grid), dim3(kWave * kRoles), 0, stream, c, a);
    return hipGetLastError();
}

hipError_t launch_int16(const ScaleArgs &s, uint32_t nvoices, hipStream_t stream)
{
    if (nvoices == 0) return hipSuccess;
    hipLaunchKernelGGL(trm_int16_kernel, dim3(nvoices), dim3(256), 0, stream, s);
    return hipGetLastError();
}

}  // namespace trm
