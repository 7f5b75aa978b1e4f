// trm_kernels.hip -- CDNA4 (gfx950) kernels of the Tube Resonance Model.
//
//   trm_noise_kernel      the voice-independent noise sequence (TRMUtility.m:71-85 + TRMFilters.m:81-86),
//                         fp64 serial recurrence, one lane, run once per batch object and cached
//   trm_tube_kernel       -[TRMTubeModel synthesize] (TRMTubeModel.m:272-361): one tube per lane,
//                         one wave (64 voices) per workgroup; state in VGPRs; wave-uniform control
//                         in SGPRs; converter coefficients + noise prefetched into LDS rings by
//                         LDS-DMA one half ahead; output staged through LDS, written as 256-byte rows
//   trm_int16_kernel      output normalisation (TRMTubeModel.m:370-389, 420-484)
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "trm_kernels.h"
#include "trm_lane.h"

namespace trm {

constexpr int kWave = 64;
constexpr int kTile = 64;            // outputs staged per lane before a flush
constexpr int kTileStride = kTile + 1;   // odd stride: conflict-free column writes and row reads

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

constexpr int kRowSlots = 64;        // converter coefficient ring: one slot per output sample
constexpr int kRowHalf = 32;         //   refilled by halves, one half ahead
constexpr int kSlotFloats = 32;      //   slot = left-wing row (16 floats) + right-wing row (16 floats)
constexpr int kNoiseRing = 128;      // noise ring: one float per tube sample, refilled by halves of 64
constexpr int kNoiseHalf = 64;

__global__ __launch_bounds__(kWave) void trm_tube_kernel(const Const C, const TubeArgs A)
{
    __shared__ float sStage[kWave * kTileStride];
    __shared__ __attribute__((aligned(16))) float sRows[kRowSlots * kSlotFloats];
    __shared__ float sNoise[kNoiseRing];
    __shared__ float sSine[kTableLen];

    const int lane = threadIdx.x;
    const uint32_t vRaw = blockIdx.x * kWave + lane;
    const bool laneValid = vRaw < A.nvoices;
    const uint32_t v = laneValid ? vRaw : A.nvoices - 1;

    if (C.waveform != 0) {
        for (int i = lane; i < kTableLen; i += kWave) sSine[i] = A.sine[i];
        __syncthreads();
    }

    const uint32_t nfr = A.nframes[v];
    const uint32_t nfrMax = wave_max_u32(nfr);
    // a voice without frames (a silent no-op, TRMTubeModel.m:274-277) reads row 0 of the buffer
    const float *frames = A.frames + (nfr > 0 ? A.frame_offset[v] * 16 : 0);
    float *const outBase = A.out + A.out_offset[v];

    const uint32_t CP = (uint32_t)C.controlPeriod;
    const uint32_t inc = C.timeRegisterIncrement;
    const uint32_t ntubeLane = nfr > 0 ? (nfr - 1) * CP : 0;
    uint32_t noutLane = 0;
    if (nfr > 0) {
        uint64_t total = (uint64_t)ntubeLane + 2ull * (uint32_t)C.padSize;
        noutLane = (uint32_t)((total * 65536ull + inc - 1) / inc);
    }
    if (!laneValid) noutLane = 0;

    Lane L;
    Track T;
    lane_reset(L);
    auto sine = [&](int i) { return sSine[i]; };

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
    auto fill_noise_half = [&](uint32_t nFirst, int half) {
        dma4(A.lp_noise + nFirst + lane, &sNoise[half * kNoiseHalf]);
    };

    if (nfrMax > 0) {
        // One flat, wave-uniform sample loop: (nfrMax-1) control periods, then the converter's
        // 2*pad zero flush (TRMRingBuffer.m:85-93).  Lanes whose utterance is shorter than the
        // wave's longest keep stepping on their last frame; their converter input is forced to 0
        // and their stores are masked by their own output count.
        const uint32_t ntubeMax = (nfrMax - 1) * CP;
        const uint32_t nTotal = ntubeMax + 2u * (uint32_t)C.padSize;
        uint32_t e = 0;          // converter read position, in pushed samples (uniform)
        uint32_t t = 0;          // 16.16 time register, N part cleared (uniform)
        uint32_t kout = 0;       // outputs emitted (uniform)
        uint32_t tilePos = 0;    // outputs staged in LDS (uniform)
        uint32_t j = CP;         // position in the control period (uniform)
        uint32_t f = 0;          // index of the period's target frame (uniform)
        float cur[16];
        {
            const float4 *p = reinterpret_cast<const float4 *>(frames);
            for (int q = 0; q < 4; q++) {
                float4 x = p[q];
                cur[4 * q] = x.x; cur[4 * q + 1] = x.y; cur[4 * q + 2] = x.z; cur[4 * q + 3] = x.w;
            }
        }
        fill_rows_half(0, 0);
        fill_rows_half(kRowHalf, 1);
        fill_noise_half(0, 0);
        fill_noise_half(kNoiseHalf, 1);
        dma_wait_all();

        for (uint32_t n = 0; n < nTotal; n++) {
            if (j == CP) {   // -setControlRateParameters:previous: (TRMTubeModel.m:289)
                j = 0;
                f++;
                float prev[16];
                for (int q = 0; q < 16; q++) prev[q] = cur[q];
                uint32_t fi = f < nfr ? f : (nfr > 0 ? nfr - 1 : 0);   // clamp: never past the voice's own rows
                const float4 *p = reinterpret_cast<const float4 *>(frames + (size_t)fi * 16);
                for (int q = 0; q < 4; q++) {
                    float4 x = p[q];
                    cur[4 * q] = x.x; cur[4 * q + 1] = x.y; cur[4 * q + 2] = x.z; cur[4 * q + 3] = x.w;
                }
                track_setup(T, C, prev, cur);
            }
            if ((n & (kNoiseHalf - 1)) == 0 && n > 0) {
                // entering a noise half: its samples were requested one half ago; refill the other half
                dma_wait_all();
                fill_noise_half(n + kNoiseHalf, ((n / kNoiseHalf) + 1) & 1);
            }
            float s = lane_sample(L, T, C, (int)j, sNoise[n & (kNoiseRing - 1)], sine);
            j++;
            s = n < ntubeLane ? s : 0.0f;
            src_push(L, s);
            while (e <= n) {     // TRMSampleRateConverter.m:171-233, uniform trip count
                if ((kout & (kRowHalf - 1)) == 0 && kout > 0) {
                    dma_wait_all();
                    fill_rows_half(kout + kRowHalf, ((kout / kRowHalf) + 1) & 1);
                }
                const float4 *rp = reinterpret_cast<const float4 *>(&sRows[(kout & (kRowSlots - 1)) * kSlotFloats]);
                float cl[16], cr[16];
                for (int q = 0; q < 4; q++) {
                    float4 a = rp[q], b = rp[4 + q];
                    cl[4 * q] = a.x; cl[4 * q + 1] = a.y; cl[4 * q + 2] = a.z; cl[4 * q + 3] = a.w;
                    cr[4 * q] = b.x; cr[4 * q + 1] = b.y; cr[4 * q + 2] = b.z; cr[4 * q + 3] = b.w;
                }
                float y = src_emit_up(L, cl, cr);
                float a = fabsf(y);
                L.maxAbs = (kout < noutLane && a > L.maxAbs) ? a : L.maxAbs;
                sStage[lane * kTileStride + tilePos] = y;
                tilePos++;
                kout++;
                t += inc;
                e += t >> 16;
                t &= 0xFFFFu;
                if (tilePos == kTile || (e > n && n + 1 == nTotal)) {
                    // flush the staged tile: row r = voice of lane r, 256 contiguous bytes per row
                    const uint32_t kbase = kout - tilePos;
#pragma unroll 1
                    for (int r = 0; r < kWave; r++) {
                        uint32_t lo = __builtin_amdgcn_readlane((uint32_t)(uintptr_t)outBase, r);
                        uint32_t hi = __builtin_amdgcn_readlane((uint32_t)((uintptr_t)outBase >> 32), r);
                        uint32_t nr = __builtin_amdgcn_readlane(noutLane, r);
                        float *dst = reinterpret_cast<float *>(((uintptr_t)hi << 32) | lo);
                        float val = sStage[r * kTileStride + lane];
                        uint32_t k = kbase + (uint32_t)lane;
                        if ((uint32_t)lane < tilePos && k < nr) dst[k] = val;
                    }
                    tilePos = 0;
                }
            }
        }
        dma_wait_all();   // nothing may still be writing LDS when the wave ends
    }

    if (laneValid) {
        A.number_samples[vRaw] = noutLane;
        A.max_sample[vRaw] = L.maxAbs;
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
    hipLaunchKernelGGL(trm_tube_kernel, dim3(grid), dim3(kWave), 0, stream, c, a);
    return hipGetLastError();
}

hipError_t launch_int16(const ScaleArgs &s, uint32_t nvoices, hipStream_t stream)
{
    if (nvoices == 0) return hipSuccess;
    hipLaunchKernelGGL(trm_int16_kernel, dim3(nvoices), dim3(256), 0, stream, s);
    return hipGetLastError();
}

}  // namespace trm
