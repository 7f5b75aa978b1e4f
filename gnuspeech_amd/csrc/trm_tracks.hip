// trm_tracks.hip -- control-track generation at 250 Hz on the device:
// -[EventList generateOutputInTimeRange:forSynthesizer:parameterLogger:] (Frameworks/GnuSpeech/MonetModel/
// EventList.m:883-1061) with MMDriftGenerator -generateDrift (MMDriftGenerator.m:65-78).
//
// One wave per utterance, lane j = value index j of the event records (0..15 the tube parameters, 16..31
// their special-event offsets, 32 the intonation contour, 33..35 the smooth-intonation slopes).  The time
// loop is the reference's (one frame per 4 ms, one event advance per frame at most); the wave-uniform part
// (time, event index, frame count) lives in SGPRs, the per-value part (current value, delta: fp64, repeated
// addition exactly as the reference does it) in the lanes.  Frames are written where the tube kernels read
// them, so a batch goes from event lists to PCM without the frames crossing PCIe.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "trm_devutil.h"
#include "trm_kernels.h"

namespace trm {

__global__ __launch_bounds__(kWave) void trm_tracks_kernel(const TrackArgs A)
{
    // every float expression below rounds per operation, like the reference's (no fused multiply-add)
#pragma clang fp contract(off)
    const uint32_t v = blockIdx.x;
    const int lane = threadIdx.x;
    const uint32_t n = A.nevents[v];
    const uint32_t *times = A.event_times + A.event_offset[v];
    const double *values = A.event_values + A.event_offset[v] * TRM_EVENT_VALUES;
    float *frames = A.frames + A.frame_offset[v] * 16;
    const trm_intonation s = A.settings;
    if (n < 2) {                                            // the reference indexes event 1 (EventList.m:920)
        if (lane == 0) A.nframes_out[v] = 0;
        return;
    }
    const int j = lane < TRM_EVENT_VALUES ? lane : TRM_EVENT_VALUES - 1;       // lanes 36..63 shadow value 35
    auto val = [&](uint32_t e) { return values[(size_t)e * TRM_EVENT_VALUES + j]; };
    uint64_t startTime = s.startTime_ms, endTime = s.endTime_ms;
    if (startTime == 0 && endTime == 0) endTime = ~0ull;    // :892-894

    // MMDriftGenerator -init / -configureWithDeviation:sampleRate:lowpassCutoff: (MMDriftGenerator.m:27-58)
    // (the seed belongs to the EventList, not to the utterance: driftSeed carries it over, MMDriftGenerator.m:41-58)
    float dPitchDeviation = 0.f, dPitchOffset = 0.f, dA0 = 0.f, dB1 = 0.f, dSeed = s.driftSeed != 0.0f ? s.driftSeed : 0.7892347f, dPrev = 0.f;
    if (s.useDrift) {                                       // :901-905
        const float sampleRate = (float)(1000u / (s.timeQuantization ? s.timeQuantization : 4u));
        float cutoff = s.driftCutoff;
        dPitchDeviation = (float)((double)s.driftDeviation * 2.0);
        dPitchOffset = s.driftDeviation;
        if (cutoff < 0.0f) cutoff = 0.0f;
        else if ((double)cutoff > ((double)sampleRate / 2.0)) cutoff = (float)((double)sampleRate / 2.0);
        dA0 = (float)(((double)cutoff * 2.0) / (double)sampleRate);
        dB1 = (float)(1.0 - (double)dA0);
    }

    // ---- starting values and deltas (:918-959)
    double cv = 0.0, cd = 0.0;
    if (j < 16) {
        uint32_t k = 1;
        double temp = val(1);
        while (isnan(temp) && ++k < n) temp = val(k);
        cv = val(0);
        cd = k < n ? ((temp - cv) / (double)times[k]) * 4.0 : 0.0;
    } else if (j == 32) {
        if (s.useSmoothIntonation) {                        // :931-941: the first contour value, no delta
            uint32_t k = 0;
            double temp = val(0);
            while (isnan(temp) && ++k < n) temp = val(k);
            cv = k < n ? temp : __builtin_nan("");
        } else {                                            // :942-959
            uint32_t k = 1;
            double temp = val(1);
            while (isnan(temp) && ++k < n) temp = val(k);
            cv = val(0);
            cd = k < n ? ((temp - cv) / (double)times[k]) * 4.0 : 0.0;
            cv = -20.0;
        }
    }

    uint32_t i = 1, count = 0;                              // :965-968
    uint64_t currentTime = 0, nextTime = times[1];
    while (i < n) {                                         // :970
        // ---- one frame (:971-1006)
        const double cvHi = __shfl(cv, (lane + 16) & 63, kWave);
        const double cv32 = __shfl(cv, 32, kWave);
        float t = (float)cv + (float)cvHi;
        {
            float t0 = t;
            if (!s.useMicroIntonation) t0 = 0.0f;
            if (s.useDrift) {                               // MMDriftGenerator.m:65-78 (uniform: every lane runs it)
                float temp = dSeed * 377.0f;
                dSeed = temp - (float)(int32_t)temp;
                temp = (dSeed * dPitchDeviation) - dPitchOffset;
                dPrev = (dA0 * temp) + (dB1 * dPrev);
                t0 += dPrev;
            }
            if (s.useMacroIntonation) t0 = (float)((double)t0 + cv32);
            t0 = (float)((double)t0 + s.pitchMean);
            if (lane == 0) t = t0;
        }
        if (currentTime >= startTime && currentTime <= endTime) {
            if (lane < 16) frames[(size_t)count * 16 + lane] = t;
            count++;
        }
        // ---- advance the values (:1008-1020)
        if (j < 32 && cd != 0.0) cv += cd;
        if (s.useSmoothIntonation) {
            const double c35 = __shfl(cd, 35, kWave);
            if (lane == 34) cd += c35;
            const double c34 = __shfl(cd, 34, kWave);
            if (lane == 33) cd += c34;
            const double c33 = __shfl(cd, 33, kWave);
            if (lane == 32) cv += c33;
        } else if (lane == 32 && cd != 0.0) {
            cv += cd;
        }
        currentTime += 4;
        // ---- next event (:1022-1054)
        if (currentTime >= nextTime) {
            i++;
            if (i == n) break;
            nextTime = times[i];
            if (j < 33 && !isnan(val(i - 1))) {
                uint32_t k = i;
                double temp = val(k);
                bool found = true;
                while (isnan(temp)) {
                    if (k >= n - 1) { cd = 0.0; found = false; break; }
                    k++;
                    temp = val(k);
                }
                if (found) cd = (temp - cv) / (double)((uint64_t)times[k] - currentTime) * 4.0;
            }
            if (s.useSmoothIntonation) {
                const double v33 = values[(size_t)(i - 1) * TRM_EVENT_VALUES + 33];
                if (!isnan(v33)) {
                    if (lane == 32) { cv = val(i - 1); cd = 0.0; }
                    if (lane >= 33 && lane < 36) cd = val(i - 1);
                }
            }
        }
    }
    if (lane == 0) A.nframes_out[v] = count;
}

hipError_t launch_tracks(const TrackArgs &a, hipStream_t stream)
{
    if (a.nvoices == 0) return hipSuccess;
    hipLaunchKernelGGL(trm_tracks_kernel, dim3(a.nvoices), dim3(kWave), 0, stream, a);
    return hipGetLastError();
}

}  // namespace trm
