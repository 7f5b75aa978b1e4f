/*
 * trm_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE (see trm_oracle.h).
 *
 * Double-precision restatement of Frameworks/Tube.  Each function cites the reference
 * lines it follows (paths relative to the GnuSpeech tree).  Arithmetic order is kept
 * where it affects rounding; containers (NSArray, NSOutputStream, Obj-C objects) are
 * replaced by one plain struct per tube.
 */
#include "trm_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

/* ---------------------------------------------------------------- constants */
enum { N_SECTIONS = 10, N_NASAL = 6, N_TAPS_FRIC = 8, WT_LEN = 512, FIR_LIMIT = 200,
       RING = 1024, ZERO_CROSSINGS = 13, L_RANGE = 256, SRC_LEN = ZERO_CROSSINGS * L_RANGE };
#define VT_SCALE 0.125                       /* TRMTubeModel.m:72 */
#define FIR_BETA 0.2                         /* TRMFIRFilter.h:7-9 */
#define FIR_GAMMA 0.1
#define FIR_CUTOFF 0.00000001
#define KAISER_BETA 5.658                    /* TRMSampleRateConverter.m:37 */
#define LP_CUTOFF (11.0 / 13.0)              /* TRMSampleRateConverter.m:11 */

/* ---------------------------------------------------------------- utility (TRMUtility.m) */
static double speed_of_sound(double t) { return 331.4 + (0.6 * t); }           /* :20-23 */

double trm_oracle_amplitude(double db)                                          /* :26-41 */
{
    db -= 60.0;
    if (db <= -60.0) return 0.0;
    if (db >= 0.0) return 1.0;
    return pow(10.0, db / 20.0);
}

double trm_oracle_frequency(double pitch)                                       /* :44-47 */
{
    return 220.0 * pow(2.0, (pitch + 3) / 12.0);
}

static double izero(double x)                                                   /* :50-66 */
{
    double sum = 1, u = 1, n = 1, halfx = x / 2.0;
    do {
        double t = halfx / n;
        n += 1;
        t *= t;
        u *= t;
        sum += u;
    } while (u >= (1E-21 * sum));
    return sum;
}

/* ---------------------------------------------------------------- FIR design (TRMFIRFilter.m) */
static void rational_approximation(double number, int *order, int *numerator, int *denominator) /* :265-310 */
{
    if (*order <= 0) { *numerator = 0; *denominator = 0; *order = -1; return; }
    double frac = fabs(number - (int)number);
    int order_max = 2 * (*order);
    if (order_max > FIR_LIMIT) order_max = FIR_LIMIT;
    int modulus = 0;
    double min_err = 1.0;
    for (int i = *order; i <= order_max; i++) {
        double ps = i * frac;
        int ip = (int)(ps + 0.5);
        double err = fabs((ps - (double)ip) / (double)i);
        if (err < min_err) { min_err = err; modulus = ip; *denominator = i; }
    }
    *numerator = (int)fabs(number) * (*denominator) + modulus;
    if (number < 0) *numerator *= -1;
    *order = *denominator - 1;
    if (*numerator == *denominator) {
        *denominator = order_max;
        *order = *numerator = *denominator - 1;
    }
}

static int maximally_flat(double beta, double gamma, int *np, double *coef)    /* :161-233 */
{
    double a[FIR_LIMIT + 1], c[FIR_LIMIT + 1];
    *np = 0;
    if (beta <= 0.0 || beta >= 0.5) return 1;
    double beta_min = ((2.0 * beta) < (1.0 - 2.0 * beta)) ? (2.0 * beta) : (1.0 - 2.0 * beta);
    if (gamma <= 0.0 || gamma >= beta_min) return 2;
    int nt = (int)(1.0 / (4.0 * gamma * gamma));
    if (nt > 160) return 3;
    double ac = (1.0 + cos(2.0 * M_PI * beta)) / 2.0;
    int numerator;
    rational_approximation(ac, &nt, &numerator, np);
    int n = (2 * (*np)) - 1;
    if (numerator == 0) numerator = 1;
    c[1] = a[1] = 1.0;
    int ll = nt - numerator;
    for (int i = 2; i <= *np; i++) {
        double sum = 1.0;
        c[i] = cos(2.0 * M_PI * ((double)(i - 1) / (double)n));
        double x = (1.0 - c[i]) / 2.0;
        double y = x;
        if (numerator == nt) continue;
        for (int j = 1; j <= ll; j++) {
            double z = y;
            if (numerator != 1)
                for (int jj = 1; jj <= numerator - 1; jj++) z *= 1.0 + ((double)j / (double)jj);
            y *= x;
            sum += z;
        }
        a[i] = sum * pow(1.0 - x, numerator);
    }
    for (int i = 1; i <= *np; i++) {
        coef[i] = a[1] / 2.0;
        for (int j = 2; j <= *np; j++) {
            int m = ((i - 1) * (j - 1)) % n;
            if (m > nt) m = n - m;
            coef[i] += c[m + 1] * a[j];
        }
        coef[i] *= 2.0 / (double)n;
    }
    return 0;
}

int trm_oracle_fir_taps(double beta, double gamma, double cutoff, double *taps, int cap) /* :37-98 */
{
    int ncoef;
    double coef[FIR_LIMIT + 1];
    memset(coef, 0, sizeof coef);
    if (maximally_flat(beta, gamma, &ncoef, coef) != 0) return -1;
    for (int i = ncoef; i > 0; i--)                                             /* trim :236-244 */
        if (fabs(coef[i]) >= fabs(cutoff)) { ncoef = i; break; }
    int ntaps = ncoef * 2 - 1;
    if (ntaps > cap) return -ntaps;
    int inc = -1, ptr = ncoef;
    for (int i = 0; i < ntaps; i++) {
        taps[i] = coef[ptr];
        ptr += inc;
        if (ptr <= 0) { ptr = 2; inc = 1; }
    }
    return ntaps;
}

/* ---------------------------------------------------------------- SRC tables */
void trm_oracle_src_tables(double *h, double *dh)               /* TRMSampleRateConverter.m:110-131 */
{
    h[0] = LP_CUTOFF;
    double x = M_PI / (double)L_RANGE;
    for (int i = 1; i < SRC_LEN; i++) {
        double y = (double)i * x;
        h[i] = sin(y * LP_CUTOFF) / y;
    }
    double ibeta = 1.0 / izero(KAISER_BETA);
    for (int i = 0; i < SRC_LEN; i++) {
        double t = (double)i / SRC_LEN;
        h[i] *= izero(KAISER_BETA * sqrt(1.0 - (t * t))) * ibeta;
    }
    for (int i = 0; i < SRC_LEN - 1; i++) dh[i] = h[i + 1] - h[i];
    dh[SRC_LEN - 1] = 0.0 - h[SRC_LEN - 1];
}

/* ---------------------------------------------------------------- per-tube state */
typedef struct {
    const trm_input_params *p;
    int32_t controlPeriod, sampleRate;
    double dampingFactor, crossmixFactor, breathinessFactor;
    /* noise (TRMUtility.m:71-85) + one-zero LP (TRMFilters.m:81-86) */
    double seed, noiseX;
    /* mouth / nose reflection-radiation pairs (TRMFilters.m:34-60) */
    double m_a10, m_b11, m_a20, m_a21, m_b21, m_reflY, m_radX, m_radY;
    double n_a10, n_b11, n_a20, n_a21, n_b21, n_reflY, n_radX, n_radY;
    /* throat (TRMFilters.m:64-77) */
    double ta0, tb1, throatY, throatGain;
    /* frication band-pass (TRMFilters.m:9-29) */
    double bpAlpha, bpBeta, bpGamma, xn1, xn2, yn1, yn2;
    /* tube memory (TRMTubeModel.m:161-172) */
    double oro[N_SECTIONS][2][2], oro_k[8];
    double nas[N_NASAL][2][2], nas_k[N_NASAL];
    double alpha[3];
    int cur, prev;
    double fricTap[N_TAPS_FRIC];
    double current[16], delta[16];
    /* wavetable (TRMWavetable.m) */
    double wavetable[WT_LEN];
    int32_t tableDiv1, tableDiv2;
    double tnLength, tnDelta, basicIncrement, currentPosition;
    /* oscillator FIR (TRMFIRFilter.m) */
    double firData[2 * FIR_LIMIT + 1], firCoef[2 * FIR_LIMIT + 1];
    int32_t firPtr, numberTaps;
    /* sample-rate converter (TRMSampleRateConverter.m) + ring buffer (TRMRingBuffer.m) */
    double sampleRateRatio, h[SRC_LEN], dh[SRC_LEN];
    uint32_t timeRegisterIncrement, filterIncrement, phaseIncrement, timeRegister;
    double ring[RING];
    int32_t padSize, fillSize, fillPtr, emptyPtr, fillCounter;
    double maximumSampleValue;
    int32_t numberSamples;
    double *out; size_t out_cap;
    double *tube; size_t tube_cap; int32_t ntube; int keep_tube;
    int oom;
} tube_t;

/* frame column indices (TRMDataList.m:223-233) */
enum { F_PITCH = 0, F_GLOTVOL, F_ASPVOL, F_FRICVOL, F_FRICPOS, F_FRICCF, F_FRICBW, F_R1, F_VELUM = 15 };

static void rr_filter_init(double coeff, double *a10, double *b11, double *a20, double *a21, double *b21)
{                                                                               /* TRMFilters.m:34-45 */
    *b11 = -coeff;
    *a10 = 1.0 - fabs(*b11);
    *a20 = coeff;
    *a21 = *b21 = -(*a20);
}

/* ---- sample-rate converter: -processDataFromRingBuffer: (TRMSampleRateConverter.m:155-298) */
static void out_push(tube_t *t, double v)
{
    if ((size_t)t->numberSamples >= t->out_cap) {
        size_t nc = t->out_cap ? t->out_cap * 2 : 65536;
        double *nb = (double *)realloc(t->out, nc * sizeof(double));
        if (!nb) { t->oom = 1; return; }
        t->out = nb; t->out_cap = nc;
    }
    t->out[t->numberSamples] = v;
}

static void ring_inc(int32_t *i) { if (++(*i) >= RING) *i -= RING; }             /* TRMRingBuffer.m:95-99 */
static void ring_dec(int32_t *i) { if (--(*i) < 0) *i += RING; }                 /* :101-105 */

#define N_MASK 0xFFFF0000u
#define nValue(x) (((x) & N_MASK) >> 16)
#define lValue(x) (((x) & 0x0000FF00u) >> 8)
#define mValue(x) ((x) & 0x000000FFu)
#define fractionValue(x) ((x) & 0x0000FFFFu)

static void data_empty(tube_t *t)
{
    int32_t endPtr = t->fillPtr - t->padSize;
    if (endPtr < 0) endPtr += RING;
    if (endPtr < t->emptyPtr) endPtr += RING;

    if (t->sampleRateRatio >= 1.0) {                                            /* :171-233 */
        while (t->emptyPtr < endPtr) {
            double output = 0.0;
            double interpolation = (double)mValue(t->timeRegister) / 256.0;
            int32_t index = t->emptyPtr;
            for (uint32_t fi = lValue(t->timeRegister); fi < SRC_LEN; ring_dec(&index), fi += t->filterIncrement)
                output += t->ring[index] * (t->h[fi] + t->dh[fi] * interpolation);
            t->timeRegister = ~t->timeRegister;
            interpolation = (double)mValue(t->timeRegister) / 256.0;
            index = t->emptyPtr;
            ring_inc(&index);
            for (uint32_t fi = lValue(t->timeRegister); fi < SRC_LEN; ring_inc(&index), fi += t->filterIncrement)
                output += t->ring[index] * (t->h[fi] + t->dh[fi] * interpolation);
            double a = fabs(output);
            if (a > t->maximumSampleValue) t->maximumSampleValue = a;
            out_push(t, output);
            t->numberSamples++;
            t->timeRegister = ~t->timeRegister;
            t->timeRegister += t->timeRegisterIncrement;
            t->emptyPtr += nValue(t->timeRegister);
            if (t->emptyPtr >= RING) { t->emptyPtr -= RING; endPtr -= RING; }
            t->timeRegister &= ~N_MASK;
        }
    } else {                                                                    /* :234-297 */
        while (t->emptyPtr < endPtr) {
            double output = 0.0, impulse;
            uint32_t phaseIndex, impulseIndex;
            phaseIndex = (uint32_t)rint(((double)fractionValue(t->timeRegister)) * t->sampleRateRatio);
            int32_t index = t->emptyPtr;
            while ((impulseIndex = (phaseIndex >> 8)) < SRC_LEN) {
                impulse = t->h[impulseIndex] + (t->dh[impulseIndex] * (((double)mValue(phaseIndex)) / 256.0));
                output += t->ring[index] * impulse;
                ring_dec(&index);
                phaseIndex += t->phaseIncrement;
            }
            phaseIndex = (uint32_t)rint(((double)fractionValue(~t->timeRegister)) * t->sampleRateRatio);
            index = t->emptyPtr;
            ring_inc(&index);
            while ((impulseIndex = (phaseIndex >> 8)) < SRC_LEN) {
                impulse = t->h[impulseIndex] + (t->dh[impulseIndex] * (((double)mValue(phaseIndex)) / 256.0));
                output += t->ring[index] * impulse;
                ring_inc(&index);
                phaseIndex += t->phaseIncrement;
            }
            double a = fabs(output);
            if (a > t->maximumSampleValue) t->maximumSampleValue = a;
            out_push(t, output);
            t->numberSamples++;
            t->timeRegister += t->timeRegisterIncrement;
            t->emptyPtr += nValue(t->timeRegister);
            if (t->emptyPtr >= RING) { t->emptyPtr -= RING; endPtr -= RING; }
            t->timeRegister &= ~N_MASK;
        }
    }
}

static void data_fill(tube_t *t, double v)                                      /* TRMRingBuffer.m:47-60 */
{
    t->ring[t->fillPtr] = v;
    ring_inc(&t->fillPtr);
    if (++t->fillCounter >= t->fillSize) { data_empty(t); t->fillCounter = 0; }
}

static void ring_flush(tube_t *t)                                               /* TRMRingBuffer.m:85-93 */
{
    for (int32_t i = 0; i < t->padSize * 2; i++) data_fill(t, 0.0);
    data_empty(t);
}

/* ---- derived constants shared by init and trm_oracle_derive */
static int derive(const trm_input_params *p, trm_derived *d)
{
    memset(d, 0, sizeof *d);
    if (!(p->length > 0.0)) return TRM_EINVAL_LENGTH;                           /* TRMTubeModel.m:197,204-207 */
    double c = speed_of_sound(p->temperature);
    d->controlPeriod = (int32_t)rint((c * N_SECTIONS * 100.0) / (p->length * p->controlRate)); /* :200 */
    d->sampleRate = (int32_t)(p->controlRate * d->controlPeriod);               /* :201 float x int32 -> int32 */
    d->actualTubeLength = (c * N_SECTIONS * 100.0) / d->sampleRate;             /* :202 */
    /* TRMSampleRateConverter.m:80-96 */
    d->sampleRateRatio = (double)p->outputRate / (double)d->sampleRate;
    d->timeRegisterIncrement = (uint32_t)(int)rint(pow(2.0, 16) / d->sampleRateRatio);
    double rounded = pow(2.0, 16) / (double)d->timeRegisterIncrement;
    if (d->sampleRateRatio >= 1.0) {
        d->phaseIncrement = 0;
        d->padSize = ZERO_CROSSINGS;
    } else {
        d->phaseIncrement = (uint32_t)rint(d->sampleRateRatio * 65536.0);
        d->padSize = (int32_t)((float)ZERO_CROSSINGS / rounded) + 1;
    }
    return TRM_OK;
}

int trm_oracle_derive(const trm_input_params *p, trm_derived *d)
{
    int rc = derive(p, d);
    if (rc) return rc;
    double taps[2 * FIR_LIMIT + 1];
    int n = trm_oracle_fir_taps(FIR_BETA, FIR_GAMMA, FIR_CUTOFF, taps, 2 * FIR_LIMIT + 1);
    if (n < 0) return TRM_EFIR;
    d->firTaps = n;
    return TRM_OK;
}

/* ---- -initWithInputData: (TRMTubeModel.m:186-260) */
static int tube_init(tube_t *t, const trm_input_params *p, trm_derived *d)
{
    memset(t, 0, sizeof *t);
    t->p = p;
    int rc = derive(p, d);
    if (rc) return rc;
    t->controlPeriod = d->controlPeriod;
    t->sampleRate = d->sampleRate;
    double nyquist = (double)t->sampleRate / 2.0;
    t->breathinessFactor = p->breathiness / 100.0;                              /* :210 */
    t->crossmixFactor = 1.0 / trm_oracle_amplitude(p->mixOffset);               /* :213 */
    t->dampingFactor = 1.0 - (p->lossFactor / 100.0);                           /* :216 */

    /* TRMWavetable -initWithWaveform:... (TRMWavetable.m:56-105) */
    t->numberTaps = trm_oracle_fir_taps(FIR_BETA, FIR_GAMMA, FIR_CUTOFF, t->firCoef, 2 * FIR_LIMIT + 1);
    if (t->numberTaps < 0) return TRM_EFIR;
    d->firTaps = t->numberTaps;
    t->firPtr = 0;
    t->tableDiv1 = (int32_t)rint(WT_LEN * (p->tp / 100.0));
    t->tableDiv2 = (int32_t)rint(WT_LEN * ((p->tp + p->tnMax) / 100.0));
    /* (the reference does not check: tp + tnMax above 100 % makes it write past its 512-entry table, TRMWavetable.m:86-96;
       the library refuses such parameters with TRM_ERANGE, and so does this restatement) */
    if (t->tableDiv1 < 0 || t->tableDiv2 > WT_LEN || t->tableDiv1 > t->tableDiv2) return TRM_ERANGE;
    t->tnLength = t->tableDiv2 - t->tableDiv1;
    t->tnDelta = rint(WT_LEN * ((p->tnMax - p->tnMin) / 100.0));
    t->basicIncrement = (double)WT_LEN / (double)t->sampleRate;
    t->currentPosition = 0;
    if (p->waveform == TRM_WAVEFORM_PULSE) {
        for (int i = 0; i < t->tableDiv1; i++) {
            double x = (double)i / (double)t->tableDiv1;
            double x2 = x * x, x3 = x2 * x;
            t->wavetable[i] = (3.0 * x2) - (2.0 * x3);
        }
        for (int i = t->tableDiv1, j = 0; i < t->tableDiv2; i++, j++) {
            double x = (double)j / t->tnLength;
            t->wavetable[i] = 1.0 - (x * x);
        }
        for (int i = t->tableDiv2; i < WT_LEN; i++) t->wavetable[i] = 0.0;
    } else {
        for (int i = 0; i < WT_LEN; i++) t->wavetable[i] = sin(((double)i / (double)WT_LEN) * 2.0 * M_PI);
    }

    rr_filter_init((nyquist - p->mouthCoef) / nyquist, &t->m_a10, &t->m_b11, &t->m_a20, &t->m_a21, &t->m_b21); /* :222 */
    rr_filter_init((nyquist - p->noseCoef) / nyquist, &t->n_a10, &t->n_b11, &t->n_a20, &t->n_a21, &t->n_b21);  /* :225 */

    /* -initializeNasalCavity (:692-707) */
    for (int i = 1, j = 1; i < 5; i++, j++) {
        double a2 = p->noseRadius[i] * p->noseRadius[i];
        double b2 = p->noseRadius[i + 1] * p->noseRadius[i + 1];
        t->nas_k[j] = (a2 - b2) / (a2 + b2);
    }
    {
        double a2 = p->noseRadius[5] * p->noseRadius[5];
        double b2 = p->apScale * p->apScale;
        t->nas_k[5] = (a2 - b2) / (a2 + b2);
    }

    t->seed = 0.7892347;                                                        /* :232, TRMUtility.m:72-77 */
    t->noiseX = 0;                                                              /* :235 */
    t->ta0 = (p->throatCutoff * 2.0) / t->sampleRate;                           /* :238, TRMFilters.m:64-68 */
    t->tb1 = 1.0 - t->ta0;
    t->throatGain = trm_oracle_amplitude(p->throatVol);                         /* :239 */

    /* TRMSampleRateConverter -initWithInputRate:outputRate: (TRMSampleRateConverter.m:69-104) */
    trm_oracle_src_tables(t->h, t->dh);
    t->sampleRateRatio = d->sampleRateRatio;
    t->timeRegisterIncrement = d->timeRegisterIncrement;
    t->filterIncrement = L_RANGE;
    t->phaseIncrement = d->phaseIncrement;
    t->timeRegister = 0;
    t->padSize = d->padSize;
    /* TRMRingBuffer -initWithPadSize: (TRMRingBuffer.m:27-44) */
    t->fillSize = RING - (2 * t->padSize);
    t->fillPtr = t->padSize;
    t->emptyPtr = 0;
    t->fillCounter = 0;

    t->cur = 1;                                                                 /* :246-247 */
    t->prev = 0;
    return TRM_OK;
}

/* -setControlRateParameters:previous: (TRMTubeModel.m:611-672), MATCH_DSP 0 */
static void set_control_rate(tube_t *t, const double *cur_in, const double *prev_in)
{
    for (int i = 0; i < 16; i++) {
        t->current[i] = prev_in[i];
        t->delta[i] = (cur_in[i] - t->current[i]) / (double)t->controlPeriod;
    }
}

/* -calculateTubeCoefficients (:712-744) */
static void tube_coefficients(tube_t *t)
{
    const double *r = &t->current[F_R1];
    for (int i = 0; i < 7; i++) {
        double a2 = r[i] * r[i], b2 = r[i + 1] * r[i + 1];
        t->oro_k[i] = (a2 - b2) / (a2 + b2);
    }
    {
        double a2 = r[7] * r[7], b2 = t->p->apScale * t->p->apScale;
        t->oro_k[7] = (a2 - b2) / (a2 + b2);
    }
    double r0_2 = r[3] * r[3], r1_2 = r0_2;
    double r2_2 = t->current[F_VELUM] * t->current[F_VELUM];
    double sum = 2.0 / (r0_2 + r1_2 + r2_2);
    t->alpha[0] = sum * r0_2;
    t->alpha[1] = sum * r1_2;
    t->alpha[2] = sum * r2_2;
    {
        double a2 = t->current[F_VELUM] * t->current[F_VELUM];
        double b2 = t->p->noseRadius[1] * t->p->noseRadius[1];
        t->nas_k[0] = (a2 - b2) / (a2 + b2);
    }
}

/* -setFricationTaps (:748-773) */
static void frication_taps(tube_t *t, int tract)
{
    double amp = trm_oracle_amplitude(t->current[F_FRICVOL]);
    if (tract) amp = 10 * amp;                      /* Applications/TRAcT/tube.c:1371 "Volume x 10 to be audible" */
    int32_t ip = (int32_t)t->current[F_FRICPOS];
    double complement = t->current[F_FRICPOS] - (double)ip;
    double remainder = 1.0 - complement;
    for (int i = 0; i < N_TAPS_FRIC; i++) {
        if (i == ip) {
            t->fricTap[i] = remainder * amp;
            if ((i + 1) < N_TAPS_FRIC) t->fricTap[++i] = complement * amp;
        } else
            t->fricTap[i] = 0.0;
    }
}

/* TRMWavetable -update: (TRMWavetable.m:117-162); scalar form (:143-149) == vDSP form */
static void wavetable_update(tube_t *t, double amp)
{
    double newDiv2 = t->tableDiv2 - rint(amp * t->tnDelta);
    double newTnLength = newDiv2 - t->tableDiv1;
    int32_t len = (int32_t)newTnLength;
    double j = 0.0;
    for (int i = 0; i < len; i++, j += 1.0) {
        double x = j / newTnLength;                 /* vDSP: j*j*(1/(L*L)); see note in DESIGN.md */
        t->wavetable[t->tableDiv1 + i] = 1.0 - (x * x);
    }
    for (int i = (int)newDiv2; i < t->tableDiv2; i++) t->wavetable[i] = 0.0;
}

static double mod0(double v) { if (v > (WT_LEN - 1)) v -= WT_LEN; return v; }   /* TRMWavetable.m:28-34 */

/* TRMFIRFilter -filterInput:needOutput: (TRMFIRFilter.m:116-146) */
static double fir_filter(tube_t *t, double in, int need)
{
    if (need) {
        double out = 0.0;
        t->firData[t->firPtr] = in;
        for (int i = 0; i < t->numberTaps; i++) {
            out += t->firData[t->firPtr] * t->firCoef[i];
            if (++t->firPtr >= t->numberTaps) t->firPtr = 0;
        }
        if (--t->firPtr < 0) t->firPtr = t->numberTaps - 1;
        return out;
    }
    t->firData[t->firPtr] = in;
    if (--t->firPtr < 0) t->firPtr = t->numberTaps - 1;
    return 0.0;
}

/* TRMWavetable -oscillator: (TRMWavetable.m:174-195), OVERSAMPLING_OSCILLATOR 1 */
static double oscillator(tube_t *t, double f)
{
    double out = 0.0;
    for (int k = 0; k < 2; k++) {
        t->currentPosition = mod0(t->currentPosition + ((f / 2.0) * t->basicIncrement));
        int32_t lo = (int32_t)t->currentPosition;
        int32_t up = (int32_t)mod0(lo + 1);
        double v = t->wavetable[lo] + ((t->currentPosition - lo) * (t->wavetable[up] - t->wavetable[lo]));
        out = fir_filter(t, v, k == 1);
    }
    return out;
}

/* -updateVocalTractWithGlottalPulse:frication: (TRMTubeModel.m:778-853) */
static double vocal_tract(tube_t *t, double input, double fric)
{
    t->cur = (t->cur + 1) % 2;
    t->prev = (t->prev + 1) % 2;
    const int c = t->cur, q = t->prev;
    const double d = t->dampingFactor;
    double (*o)[2][2] = t->oro;
    double (*n)[2][2] = t->nas;
    enum { TOP = 0, BOT = 1 };
    double delta;

    o[0][TOP][c] = (o[0][BOT][q] * d) + input;
    delta = t->oro_k[0] * (o[0][TOP][q] - o[1][BOT][q]);
    o[1][TOP][c] = (o[0][TOP][q] + delta) * d;
    o[0][BOT][c] = (o[1][BOT][q] + delta) * d;
    for (int i = 1, j = 1, k = 0; i < 3; i++, j++, k++) {
        delta = t->oro_k[j] * (o[i][TOP][q] - o[i + 1][BOT][q]);
        o[i + 1][TOP][c] = ((o[i][TOP][q] + delta) * d) + (t->fricTap[k] * fric);
        o[i][BOT][c] = (o[i + 1][BOT][q] + delta) * d;
    }
    double jp = (t->alpha[0] * o[3][TOP][q]) + (t->alpha[1] * o[4][BOT][q]) + (t->alpha[2] * n[0][BOT][q]);
    o[3][BOT][c] = (jp - o[3][TOP][q]) * d;
    o[4][TOP][c] = ((jp - o[4][BOT][q]) * d) + (t->fricTap[2] * fric);
    n[0][TOP][c] = (jp - n[0][BOT][q]) * d;
    delta = t->oro_k[3] * (o[4][TOP][q] - o[5][BOT][q]);
    o[5][TOP][c] = ((o[4][TOP][q] + delta) * d) + (t->fricTap[3] * fric);
    o[4][BOT][c] = (o[5][BOT][q] + delta) * d;
    o[6][TOP][c] = (o[5][TOP][q] * d) + (t->fricTap[4] * fric);
    o[5][BOT][c] = o[6][BOT][q] * d;
    for (int i = 6, j = 4, k = 5; i < 9; i++, j++, k++) {
        delta = t->oro_k[j] * (o[i][TOP][q] - o[i + 1][BOT][q]);
        o[i + 1][TOP][c] = ((o[i][TOP][q] + delta) * d) + (t->fricTap[k] * fric);
        o[i][BOT][c] = (o[i + 1][BOT][q] + delta) * d;
    }
    /* mouth reflection (low-pass) and radiation (high-pass), TRMFilters.m:47-60 */
    {
        double in = t->oro_k[7] * o[9][TOP][q];
        double y = (t->m_a10 * in) - (t->m_b11 * t->m_reflY);
        t->m_reflY = y;
        o[9][BOT][c] = d * y;
    }
    double output;
    {
        double in = (1.0 + t->oro_k[7]) * o[9][TOP][q];
        double y = (t->m_a20 * in) + (t->m_a21 * t->m_radX) - (t->m_b21 * t->m_radY);
        t->m_radX = in;
        t->m_radY = y;
        output = y;
    }
    for (int i = 0, j = 0; i < 5; i++, j++) {
        delta = t->nas_k[j] * (n[i][TOP][q] - n[i + 1][BOT][q]);
        n[i + 1][TOP][c] = (n[i][TOP][q] + delta) * d;
        n[i][BOT][c] = (n[i + 1][BOT][q] + delta) * d;
    }
    {
        double in = t->nas_k[5] * n[5][TOP][q];
        double y = (t->n_a10 * in) - (t->n_b11 * t->n_reflY);
        t->n_reflY = y;
        n[5][BOT][c] = d * y;
    }
    {
        double in = (1.0 + t->nas_k[5]) * n[5][TOP][q];
        double y = (t->n_a20 * in) + (t->n_a21 * t->n_radX) - (t->n_b21 * t->n_radY);
        t->n_radX = in;
        t->n_radY = y;
        output += y;
    }
    return output;
}

static void tube_push(tube_t *t, double v)
{
    if (!t->keep_tube) return;
    if ((size_t)t->ntube >= t->tube_cap) {
        size_t nc = t->tube_cap ? t->tube_cap * 2 : 32768;
        double *nb = (double *)realloc(t->tube, nc * sizeof(double));
        if (!nb) { t->oom = 1; return; }
        t->tube = nb; t->tube_cap = nc;
    }
    t->tube[t->ntube++] = v;
}

/* tract != 0: the sample loop of Applications/TRAcT/tube.c (its synthesize() thread, tube.c:1096-1190) instead of
 * Frameworks/Tube's: frame f is the parameter set `current` holds during control period f -- no interpolation
 * (tube.c:1121-1136 converts current.* every sample) --, ten times the frication amplitude (tube.c:1371), the tube-rate
 * sample times 100 before the converter (tube.c:1177).  Pinned against the reference binary run in that order
 * (oracle/ref_driver.c `tract`, tests/golden/tract_mode_*.npz). */
/* slice: TRAcT's loop order only -- the tube samples every frame is held for (0: a control period).  tube.c reads `current`
 * every sample (tube.c:1121-1136), so parameters may change at any sample; a frame per `slice` samples is that loop with
 * the changes on a grid of `slice` samples (trm_stream_set_slice). */
static int synthesize_impl(const trm_input_params *p, const double *frames, size_t nframes,
                           int keep_tube_samples, trm_oracle_result *out, int tract, int32_t slice)
{
    if (!p || !out || (nframes && !frames)) return TRM_EINVAL;
    memset(out, 0, sizeof *out);
    tube_t *t = (tube_t *)malloc(sizeof *t);
    if (!t) return TRM_ENOMEM;
    int rc = tube_init(t, p, &out->derived);
    if (rc) { free(t); return rc; }
    t->keep_tube = keep_tube_samples;

    if (nframes > 0) {                                                          /* TRMTubeModel.m:274-277 */
        for (size_t f = 1; f < nframes; f++) {                                  /* :282-357 */
            set_control_rate(t, frames + 16 * f, frames + 16 * (tract ? f : f - 1));
            const int32_t run = (tract && slice > 0) ? slice : t->controlPeriod;
            for (int32_t j = 0; j < run; j++) {
                double f0 = trm_oracle_frequency(t->current[F_PITCH]);          /* :294-296 */
                double ax = trm_oracle_amplitude(t->current[F_GLOTVOL]);
                double ah1 = trm_oracle_amplitude(t->current[F_ASPVOL]);
                tube_coefficients(t);                                           /* :298 */
                frication_taps(t, tract);                                       /* :299 */
                {                                                               /* :300, TRMFilters.m:9-17 */
                    double tanv = tan((M_PI * t->current[F_FRICBW]) / t->sampleRate);
                    double cosv = cos((2.0 * M_PI * t->current[F_FRICCF]) / t->sampleRate);
                    t->bpBeta = (1.0 - tanv) / (2.0 * (1.0 + tanv));
                    t->bpGamma = (0.5 + t->bpBeta) * cosv;
                    t->bpAlpha = (0.5 - t->bpBeta) / 2.0;
                }
                double lp_noise;                                                /* :305 */
                {
                    double product = t->seed * 377.0;
                    t->seed = product - (int)product;
                    double nz = t->seed - 0.5;
                    lp_noise = nz + t->noiseX;
                    t->noiseX = nz;
                }
                if (p->waveform == TRM_WAVEFORM_PULSE) wavetable_update(t, ax); /* :308-309 */
                double pulse = oscillator(t, f0);                               /* :312 */
                double pulsed_noise = lp_noise * pulse;                         /* :315 */
                pulse = ax * ((pulse * (1.0 - t->breathinessFactor)) + (pulsed_noise * t->breathinessFactor)); /* :318 */
                double signal;
                if (p->usesModulation) {                                        /* :323-333 */
                    double crossmix = ax * t->crossmixFactor;
                    crossmix = (crossmix < 1.0) ? crossmix : 1.0;
                    signal = (pulsed_noise * crossmix) + (lp_noise * (1.0 - crossmix));
                } else
                    signal = lp_noise;
                double bp;                                                      /* TRMFilters.m:19-29 */
                {
                    bp = 2.0 * ((t->bpAlpha * (signal - t->xn2)) + (t->bpGamma * t->yn1) - (t->bpBeta * t->yn2));
                    t->xn2 = t->xn1; t->xn1 = signal; t->yn2 = t->yn1; t->yn1 = bp;
                }
                signal = vocal_tract(t, (pulse + (ah1 * signal)) * VT_SCALE, bp); /* :336-337 */
                {                                                               /* :341, TRMFilters.m:72-77 */
                    double y = (t->ta0 * (pulse * VT_SCALE)) + (t->tb1 * t->throatY);
                    t->throatY = y;
                    signal += y * t->throatGain;
                }
                if (tract) signal = signal * 100;                               /* tube.c:1177 */
                tube_push(t, signal);
                data_fill(t, signal);                                           /* :346 */
                for (int i = 0; i < 16; i++) t->current[i] += t->delta[i];      /* :351, :676-688 */
            }
        }
        ring_flush(t);                                                          /* :360 */
    }

    rc = t->oom ? TRM_ENOMEM : TRM_OK;
    out->samples = t->out;
    out->numberSamples = t->numberSamples;
    out->maximumSampleValue = t->maximumSampleValue;
    out->tubeSamples = t->tube;
    out->numberTubeSamples = t->ntube;
    free(t);
    if (rc) trm_oracle_result_free(out);
    return rc;
}

int trm_oracle_synthesize(const trm_input_params *p, const double *frames, size_t nframes,
                          int keep_tube_samples, trm_oracle_result *out)
{
    return synthesize_impl(p, frames, nframes, keep_tube_samples, out, 0, 0);
}

int trm_oracle_synthesize_tract(const trm_input_params *p, const double *frames, size_t nframes,
                                int keep_tube_samples, trm_oracle_result *out)
{
    return synthesize_impl(p, frames, nframes, keep_tube_samples, out, 1, 0);
}

int trm_oracle_synthesize_tract_slices(const trm_input_params *p, const double *frames, size_t nframes, int32_t slice,
                                       int keep_tube_samples, trm_oracle_result *out)
{
    if (slice < 1) return TRM_EINVAL;
    return synthesize_impl(p, frames, nframes, keep_tube_samples, out, 1, slice);
}

/* bench.py's cpu_baseline leg: `count` voices of `nframes` frames each (frames = [voices][nframes][16] doubles),
 * starting at voice `first` and wrapping at `nvoices`, one after the other on the calling thread; returns the output
 * samples produced.  The library keeps no global state: any number of threads may run this at once. */
int trm_oracle_run_voices(const trm_input_params *p, const double *frames, size_t nframes, size_t nvoices,
                          size_t first, size_t count, uint64_t *samples_out)
{
    uint64_t total = 0;
    if (!p || !frames || nvoices == 0) return TRM_EINVAL;
    for (size_t i = 0; i < count; i++) {
        trm_oracle_result r;
        const size_t v = (first + i) % nvoices;
        int rc = trm_oracle_synthesize(p, frames + v * nframes * 16, nframes, 0, &r);
        if (rc) return rc;
        total += (uint64_t)r.numberSamples;
        trm_oracle_result_free(&r);
    }
    if (samples_out) *samples_out = total;
    return TRM_OK;
}

void trm_oracle_result_free(trm_oracle_result *r)
{
    if (!r) return;
    free(r->samples);
    free(r->tubeSamples);
    r->samples = r->tubeSamples = NULL;
    r->numberSamples = r->numberTubeSamples = 0;
}

void trm_oracle_lp_noise(double *lp, size_t count)
{
    double seed = 0.7892347, x1 = 0.0;
    for (size_t i = 0; i < count; i++) {
        double product = seed * 377.0;
        seed = product - (int)product;
        double nz = seed - 0.5;
        lp[i] = nz + x1;
        x1 = nz;
    }
}

/* ---------------------------------------------------------------- output writers */
/* double -> int16 as the reference's x86 build does it for out-of-range values (the stereo file
 * path over-drives a channel by up to 2x): convert to a wide integer, keep the low 16 bits. */
static int16_t wrap16(double v) { return (int16_t)(uint16_t)(int64_t)v; }

void trm_oracle_scale_int16(const trm_input_params *p, const double *s, int32_t n, double maxv,
                            int for_wav_data, int16_t *out)
{
    double scale = (32767.0 / maxv) * trm_oracle_amplitude(p->volume);          /* TRMTubeModel.m:370,515 */
    if (p->channels == 2) {
        double l, r;
        if (for_wav_data) {                                                     /* :532-533 */
            l = -((p->balance / 2.0) - 0.5) * scale;
            r = ((p->balance / 2.0) + 0.5) * scale;
        } else {                                                                /* :382-383 */
            l = -((p->balance / 2.0) - 0.5) * scale * 2.0;
            r = ((p->balance / 2.0) + 0.5) * scale * 2.0;
        }
        for (int32_t i = 0; i < n; i++) {
            out[2 * i] = wrap16(rint(s[i] * l));
            out[2 * i + 1] = wrap16(rint(s[i] * r));
        }
    } else {
        for (int32_t i = 0; i < n; i++) out[i] = wrap16(rint(s[i] * scale));
    }
}

static uint8_t *put_be32(uint8_t *b, uint32_t v) { b[0] = v >> 24; b[1] = v >> 16; b[2] = v >> 8; b[3] = v; return b + 4; }
static uint8_t *put_le32(uint8_t *b, uint32_t v) { b[0] = v; b[1] = v >> 8; b[2] = v >> 16; b[3] = v >> 24; return b + 4; }
static uint8_t *put_le16(uint8_t *b, uint16_t v) { b[0] = v & 0xff; b[1] = v >> 8; return b + 2; }

size_t trm_oracle_wav_data(const trm_input_params *p, const double *s, int32_t n, double maxv,
                           uint8_t *buf, size_t cap)                           /* TRMTubeModel.m:509-593 */
{
    int ch = p->channels == 2 ? 2 : 1;
    size_t data_bytes = (size_t)n * 2 * ch;
    size_t total = 12 + (8 + 18) + 8 + data_bytes;
    if (!buf || cap < total) return 0;
    int frameSize = (int)ceil(p->channels * (16.0 / 8));
    int bytesPerSecond = (int)ceil(p->outputRate * frameSize);
    uint8_t *b = buf;
    b = put_be32(b, 0x52494646);
    b = put_le32(b, (uint32_t)(4 + (8 + 18) + (8 + data_bytes)));
    b = put_be32(b, 0x57415645);
    b = put_be32(b, 0x666d7420);
    b = put_le32(b, 18);
    b = put_le16(b, 1);
    b = put_le16(b, (uint16_t)p->channels);
    b = put_le32(b, (uint32_t)p->outputRate);
    b = put_le32(b, (uint32_t)bytesPerSecond);
    b = put_le16(b, (uint16_t)frameSize);
    b = put_le16(b, 16);
    b = put_le16(b, 0);
    b = put_be32(b, 0x64617461);
    b = put_le32(b, (uint32_t)data_bytes);
    int16_t *pcm = (int16_t *)malloc(data_bytes ? data_bytes : 2);
    if (!pcm) return 0;
    trm_oracle_scale_int16(p, s, n, maxv, 1, pcm);
    for (size_t i = 0; i < (size_t)n * ch; i++) b = put_le16(b, (uint16_t)pcm[i]);
    free(pcm);
    return total;
}

/* ---------------------------------------------------------------- .trm parser (TRMDataList.m:43-247) */
int trm_oracle_parse_file(const char *path, trm_input_params *p, double **frames, size_t *nframes)
{
    FILE *fp = fopen(path, "r");
    if (!fp) return TRM_EIO;
    char line[128];
    memset(p, 0, sizeof *p);
    *frames = NULL; *nframes = 0;
#define NEXT() do { if (!fgets(line, 128, fp)) { fclose(fp); return TRM_EPARSE; } } while (0)
    NEXT(); p->outputFileFormat = (int32_t)strtol(line, NULL, 10);
    NEXT(); p->outputRate = (float)strtod(line, NULL);
    NEXT(); p->controlRate = (float)strtod(line, NULL);
    NEXT(); p->volume = strtod(line, NULL);
    NEXT(); p->channels = (int32_t)strtol(line, NULL, 10);
    NEXT(); p->balance = strtod(line, NULL);
    NEXT(); p->waveform = (int32_t)strtol(line, NULL, 10);
    NEXT(); p->tp = strtod(line, NULL);
    NEXT(); p->tnMin = strtod(line, NULL);
    NEXT(); p->tnMax = strtod(line, NULL);
    NEXT(); p->breathiness = strtod(line, NULL);
    NEXT(); p->length = strtod(line, NULL);
    NEXT(); p->temperature = strtod(line, NULL);
    NEXT(); p->lossFactor = strtod(line, NULL);
    NEXT(); p->apScale = strtod(line, NULL);
    NEXT(); p->mouthCoef = strtod(line, NULL);
    NEXT(); p->noseCoef = strtod(line, NULL);
    for (int i = 1; i < TRM_TOTAL_NASAL_SECTIONS; i++) { NEXT(); p->noseRadius[i] = strtod(line, NULL); }
    NEXT(); p->throatCutoff = strtod(line, NULL);
    NEXT(); p->throatVol = strtod(line, NULL);
    NEXT(); p->usesModulation = (strtol(line, NULL, 10) != 0);
    NEXT(); p->mixOffset = strtod(line, NULL);
#undef NEXT
    size_t cap = 0, n = 0;
    double *f = NULL;
    while (fgets(line, 128, fp)) {
        if (n + 2 > cap) {
            cap = cap ? cap * 2 : 512;
            double *nf = (double *)realloc(f, cap * 16 * sizeof(double));
            if (!nf) { free(f); fclose(fp); return TRM_ENOMEM; }
            f = nf;
        }
        char *ptr = line;
        for (int i = 0; i < 16; i++) f[n * 16 + i] = strtod(ptr, &ptr);
        n++;
    }
    if (n > 0) { memcpy(f + n * 16, f + (n - 1) * 16, 16 * sizeof(double)); n++; }  /* :239-241 */
    fclose(fp);
    *frames = f; *nframes = n;
    return TRM_OK;
}
