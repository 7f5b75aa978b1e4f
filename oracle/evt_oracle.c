/* evt_oracle.c -- TEST INFRASTRUCTURE (parity oracle).  CPU restatement of the reference's
 * control-track generator: -[EventList generateOutputInTimeRange:forSynthesizer:parameterLogger:]
 * (Frameworks/GnuSpeech/MonetModel/EventList.m:883-1061) and MMDriftGenerator
 * (Frameworks/GnuSpeech/MonetModel/MMDriftGenerator.m:41-78), statement by statement.
 *
 * PARITY UNPINNED: the reference holds no event-list fixture and EventList.m needs Foundation, so this
 * restatement is checked against hand-computed cases only (tests/test_events.py).  Two places where the
 * reference's behaviour is undefined are made definite here and in the HIP path alike:
 *   - a parameter with no non-NaN target after event 0 (EventList.m:919-921 walks off the array): delta 0;
 *   - MMDriftGenerator's float expressions are evaluated without fused multiply-add (x86-64 semantics;
 *     an arm64 build of the reference may contract `a0*temp + b1*prev`).
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may use anything under oracle/. */
#include <math.h>
#include <stddef.h>
#include <stdint.h>

#include "../include/trm_c_api.h"
#include "trm_oracle.h"

/* MMDriftGenerator.m:6-7,27-36 */
typedef struct {
    float pitchDeviation, pitchOffset, a0, b1, seed, previousSample;
} drift_t;

static void drift_init(drift_t *d)
{
    d->pitchDeviation = 0; d->pitchOffset = 0; d->a0 = 0; d->b1 = 0;
    d->seed = 0.7892347f;
    d->previousSample = 0.0f;
}

/* MMDriftGenerator.m:41-58 */
static void drift_configure(drift_t *d, float deviation, float sampleRate, float lowpassCutoff)
{
    d->pitchDeviation = (float)(deviation * 2.0);
    d->pitchOffset = deviation;
    if (lowpassCutoff < 0.0) lowpassCutoff = 0.0f;
    else if (lowpassCutoff > (sampleRate / 2.0)) lowpassCutoff = (float)(sampleRate / 2.0);
    d->a0 = (float)((lowpassCutoff * 2.0) / sampleRate);
    d->b1 = (float)(1.0 - d->a0);
    d->previousSample = 0.0f;
}

/* MMDriftGenerator.m:65-78 */
static float drift_generate(drift_t *d)
{
    volatile float temp = d->seed * 377.0f;            /* volatile: each float operation rounds to float */
    d->seed = temp - (float)(int32_t)temp;
    volatile float scaled = d->seed * d->pitchDeviation;
    temp = scaled - d->pitchOffset;
    volatile float p = d->a0 * temp, q = d->b1 * d->previousSample;
    d->previousSample = p + q;
    return d->previousSample;
}

#define VAL(e, i) values[(size_t)(e) * TRM_EVENT_VALUES + (i)]

int trm_oracle_count_frames(const uint32_t *times, size_t n, const trm_intonation *s, size_t *nframes)
{
    *nframes = 0;
    if (n < 2) return 0;
    uint64_t start = s->startTime_ms, end = s->endTime_ms;
    if (start == 0 && end == 0) end = UINT64_MAX;                 /* EventList.m:892-894 */
    size_t i = 1, count = 0;
    uint64_t t = 0, nextTime = times[1];
    while (i < n) {
        if (t >= start && t <= end) count++;                      /* :985 */
        t += 4;                                                   /* :1020 */
        if (t >= nextTime) {                                      /* :1022 */
            i++;
            if (i == n) break;
            nextTime = times[i];
        }
    }
    *nframes = count;
    return 0;
}

/* frames_out: room for frames_cap rows of 16 floats; *nframes = rows the generator emits (even beyond cap). */
int trm_oracle_generate_frames(const uint32_t *times, const double *values, size_t n, const trm_intonation *s,
                               float *frames_out, size_t frames_cap, size_t *nframes)
{
    *nframes = 0;
    if (n < 2) return 0;                                          /* (:889-890 returns on 0 events; event 1 is indexed at :920) */
    uint64_t startTime = s->startTime_ms, endTime = s->endTime_ms;
    if (startTime == 0 && endTime == 0) endTime = UINT64_MAX;     /* :892-894 */

    drift_t drift;
    drift_init(&drift);
    if (s->driftSeed != 0.0f) drift.seed = s->driftSeed;          /* the EventList's generator continues (MMDriftGenerator.m:41-58) */
    if (s->useDrift)                                              /* :901-905 */
        drift_configure(&drift, s->driftDeviation, (float)(1000u / (s->timeQuantization ? s->timeQuantization : 4u)), s->driftCutoff);

    const double millisecondsPerInterval = 4.0;                   /* :907-909 */
    double currentValues[36], currentDeltas[36], temp = 0.0;
    for (size_t i = 0; i < 16; i++) {                             /* :918-925 */
        size_t j = 1;
        while (j < n && isnan(temp = VAL(j, i))) j++;
        currentValues[i] = VAL(0, i);
        if (j < n)
            currentDeltas[i] = ((temp - currentValues[i]) / (double)times[j]) * millisecondsPerInterval;
        else
            currentDeltas[i] = 0.0;                               /* (undefined in the reference; see the header) */
    }
    for (size_t i = 16; i < 36; i++) currentValues[i] = currentDeltas[i] = 0.0;   /* :928-929 */

    if (s->useSmoothIntonation) {                                 /* :931-941 */
        size_t j = 0;
        while (isnan(temp = VAL(j, 32))) {
            j++;
            if (j >= n) break;
        }
        currentValues[32] = j < n ? VAL(j, 32) : NAN;             /* (the reference reads past the array when none exists) */
        currentDeltas[32] = 0.0;
    } else {                                                      /* :942-959 */
        size_t j = 1;
        while (isnan(temp = VAL(j, 32))) {
            j++;
            if (j >= n) break;
        }
        currentValues[32] = VAL(0, 32);
        if (j < n)
            currentDeltas[32] = ((temp - currentValues[32]) / (double)times[j]) * millisecondsPerInterval;
        else
            currentDeltas[32] = 0;
        currentValues[32] = -20.0;
    }

    size_t i = 1, count = 0;                                      /* :965-968 */
    uint64_t currentTime = 0, nextTime = times[1];
    float table[16];
    while (i < n) {                                               /* :970 */
        for (size_t j = 0; j < 16; j++) table[j] = (float)currentValues[j] + (float)currentValues[j + 16];   /* :971-973 */
        if (!s->useMicroIntonation) table[0] = 0.0f;              /* :974-975 */
        if (s->useDrift) table[0] += drift_generate(&drift);      /* :976-977 */
        if (s->useMacroIntonation) table[0] = (float)(table[0] + currentValues[32]);   /* :978-981 */
        table[0] = (float)(table[0] + s->pitchMean);              /* :983 */

        if (currentTime >= startTime && currentTime <= endTime) { /* :985-1006 */
            if (count < frames_cap)
                for (size_t j = 0; j < 16; j++) frames_out[count * 16 + j] = table[j];
            count++;
        }

        for (size_t j = 0; j < 32; j++)                           /* :1008-1011 */
            if (currentDeltas[j]) currentValues[j] += currentDeltas[j];
        if (s->useSmoothIntonation) {                             /* :1012-1015 */
            currentDeltas[34] += currentDeltas[35];
            currentDeltas[33] += currentDeltas[34];
            currentValues[32] += currentDeltas[33];
        } else {
            if (currentDeltas[32]) currentValues[32] += currentDeltas[32];   /* :1017-1018 */
        }
        currentTime += 4;                                         /* :1020 */

        if (currentTime >= nextTime) {                            /* :1022 */
            i++;
            if (i == n) break;
            nextTime = times[i];
            for (size_t j = 0; j < 33; j++) {                     /* :1028-1044 */
                if (!isnan(VAL(i - 1, j))) {
                    size_t k = i;
                    while (isnan(temp = VAL(k, j))) {
                        if (k >= n - 1) {
                            currentDeltas[j] = 0.0;
                            break;
                        }
                        k++;
                    }
                    if (!isnan(temp))
                        currentDeltas[j] = (temp - currentValues[j]) / (double)((uint64_t)times[k] - currentTime) * millisecondsPerInterval;
                }
            }
            if (s->useSmoothIntonation) {                         /* :1045-1053 */
                if (!isnan(VAL(i - 1, 33))) {
                    currentValues[32] = VAL(i - 1, 32);
                    currentDeltas[32] = 0.0;
                    currentDeltas[33] = VAL(i - 1, 33);
                    currentDeltas[34] = VAL(i - 1, 34);
                    currentDeltas[35] = VAL(i - 1, 35);
                }
            }
        }
    }
    *nframes = count;
    return 0;
}
