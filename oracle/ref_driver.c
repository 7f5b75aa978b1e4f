/*
 * ref_driver.c -- TEST INFRASTRUCTURE.  Drives the REFERENCE's own plain-C tube
 * (Applications/TRAcT/tube.c, compiled where it lies under /root/reference by
 * oracle/Makefile; never copied) through its per-sample functions in the order of
 * -[TRMTubeModel synthesize] (Frameworks/Tube/TRMTubeModel.m:294-351), so that
 * oracle/trm_oracle.c can be pinned against reference-computed samples.
 *
 * tube.c keeps all state in globals / function-local statics, so ONE case per process.
 *
 * Known divergences of tube.c from Frameworks/Tube neutralised here:
 *  - its real-time synthesize() thread is not started (threadFlag preset to 1) and its
 *    x100 output gain (tube.c:1177) is not applied: the driver owns the sample loop;
 *  - setFricationTaps() scales the amplitude by 10 (tube.c:1371); the driver installs
 *    the framework's taps (TRMTubeModel.m:748-773) built from tube.c's own amplitude(),
 *    and records max |ref_tap/10 - tap| as a cross-check;
 *  - control-rate interpolation is the driver's (tube.c has no frame input): repeated
 *    addition of (cur-prev)/controlPeriod exactly as TRMTubeModel.m:611-688;
 *  - dataEmpty() emits (float)output into circBuff2 (tube.c:2414-2421): the driver
 *    drains that buffer after every dataFill, so converter outputs are fp32-rounded;
 *    numberSamples and maximumSampleValue are read as the doubles/longs tube.c keeps.
 *
 * usage: tube_ref <case.bin> <out.bin> [tract [slice=N]]
 *   case.bin = trm_input_params | uint64 nframes | nframes*16 doubles
 *   out.bin  = see write_out() below
 *
 * `tract`: TRAcT's OWN sample loop instead (tube.c:1096-1190, its synthesize() thread body, one iteration per
 * sample): none of the four divergences is neutralised.  tube.c has no frame input -- its GUI writes `current`
 * whenever a slider moves -- so the driver holds frame f for the whole control period f (parameters STEP at the
 * period boundaries, no interpolation), keeps tube.c's own setFricationTaps() (x10, tube.c:1371) and multiplies the
 * tube-rate sample by 100 before dataFill (tube.c:1177).  This is the golden for shim/tract_tube.c and the streams.
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../include/trm_c_api.h"

/* ---- symbols defined by tube.c (declared here; its headers define storage) ---- */
extern float  outputRate, controlRate;
extern double volume, balance, tp, tnMin, tnMax, breathiness, length, lossFactor, apScale,
              mouthCoef, noseCoef, noseRadius[], throatCutoff, throatVol, mixOffset;
extern int    channels, waveform, modulation, controlPeriod, sampleRate;
extern double fricationTap[8];
extern double breathinessFactor, crossmixFactor;
extern double maximumSampleValue;
extern long   numberSamples;
extern double h[], deltaH[];
extern double *FIRCoef;
extern int    numberTaps, padSize;
extern unsigned int timeRegisterIncrement, phaseIncrement;
extern float  circBuff2[];
extern float *circBuff2InPtr, *circBuff2OutPtr;

void   setTemperature(double value);
int   *getThreadFlag(void);
int    initializeSynthesizer(void);
void   initCircBuff2(void);
double *getGlotPitch(void), *getGlotVol(void), *getAspVol(void), *getFricVol(void), *getFricPos(void),
       *getFricCF(void), *getFricBW(void), *getRadius(int), *getVelumRadius(void);
double amplitude(double), frequency(double), noise(void), noiseFilter(double), oscillator(double),
       vocalTract(double, double), throat(double), bandpassFilter(double);
void   calculateTubeCoefficients(void), setFricationTaps(void), calculateBandpassCoefficients(void),
       updateWavetable(double), dataFill(double), flushBuffer(void);

static float *g_out; static size_t g_nout, g_cap;
static void drain(void)
{
    /* circBuff2 is linear between resets: the driver resets it after every drain */
    size_t n = (size_t)(circBuff2InPtr - circBuff2OutPtr);
    if (g_nout + n > g_cap) {
        g_cap = (g_nout + n) * 2 + 65536;
        g_out = (float *)realloc(g_out, g_cap * sizeof(float));
        if (!g_out) { fprintf(stderr, "oom\n"); exit(2); }
    }
    memcpy(g_out + g_nout, circBuff2OutPtr, n * sizeof(float));
    g_nout += n;
    initCircBuff2();
}

int main(int argc, char **argv)
{
    if (argc < 3 || argc > 5) { fprintf(stderr, "usage: %s case.bin out.bin [tract [slice=N]]\n", argv[0]); return 2; }
    const int tract = argc >= 4 && !strcmp(argv[3], "tract");
    /* tract slice=N: every frame is held for N tube samples instead of a control period -- tube.c reads `current` every
     * sample (tube.c:1121-1136), so this is its loop with the GUI's writes on a grid of N samples */
    const int slice = (tract && argc == 5 && !strncmp(argv[4], "slice=", 6)) ? atoi(argv[4] + 6) : 0;
    FILE *fi = fopen(argv[1], "rb");
    if (!fi) { perror("case"); return 2; }
    trm_input_params p;
    uint64_t nframes;
    if (fread(&p, sizeof p, 1, fi) != 1 || fread(&nframes, 8, 1, fi) != 1) return 2;
    double *frames = (double *)malloc((nframes ? nframes : 1) * 16 * sizeof(double));
    if (fread(frames, 16 * sizeof(double), nframes, fi) != nframes) return 2;
    fclose(fi);

    /* tube.c prints diagnostics on stdout (and every down-sampled value, tube.c:2464) */
    if (!freopen("/dev/null", "w", stdout)) return 2;

    *getThreadFlag() = 1;                 /* no real-time pthread (tube.c:587,666-675) */
    outputRate = p.outputRate; controlRate = p.controlRate; volume = p.volume; channels = p.channels;
    balance = p.balance; waveform = p.waveform; tp = p.tp; tnMin = p.tnMin; tnMax = p.tnMax;
    breathiness = p.breathiness; length = p.length; setTemperature(p.temperature);
    lossFactor = p.lossFactor; apScale = p.apScale; mouthCoef = p.mouthCoef; noseCoef = p.noseCoef;
    for (int i = 0; i < 6; i++) noseRadius[i] = p.noseRadius[i];
    throatCutoff = p.throatCutoff; throatVol = p.throatVol; modulation = p.usesModulation;
    mixOffset = p.mixOffset;
    initCircBuff2();
    if (initializeSynthesizer() != 0) { fprintf(stderr, "initializeSynthesizer failed\n"); return 3; }

    const int run = slice > 0 ? slice : controlPeriod;
    size_t ntube_cap = nframes > 1 ? (size_t)(nframes - 1) * (size_t)run : 0, ntube = 0;
    double *tube = (double *)malloc((ntube_cap ? ntube_cap : 1) * sizeof(double));
    double tap_err = 0.0;
    double cur[16], delta[16];

    for (uint64_t f = 1; f < nframes; f++) {
        const double *prev_in = frames + 16 * (f - 1), *cur_in = frames + 16 * f;
        for (int i = 0; i < 16; i++) {                       /* TRMTubeModel.m:611-672 */
            cur[i] = prev_in[i];
            delta[i] = (cur_in[i] - cur[i]) / (double)controlPeriod;
            if (tract) { cur[i] = cur_in[i]; delta[i] = 0.0; }   /* TRAcT: the set the GUI left in `current`, held */
        }
        for (int j = 0; j < run; j++) {
            *getGlotPitch() = cur[0]; *getGlotVol() = cur[1]; *getAspVol() = cur[2]; *getFricVol() = cur[3];
            *getFricPos() = cur[4]; *getFricCF() = cur[5]; *getFricBW() = cur[6];
            for (int i = 0; i < 8; i++) *getRadius(i) = cur[7 + i];
            *getVelumRadius() = cur[15];

            double f0 = frequency(cur[0]);                   /* TRMTubeModel.m:294-296 */
            double ax = amplitude(cur[1]);
            double ah1 = amplitude(cur[2]);
            calculateTubeCoefficients();                     /* :298 */
            setFricationTaps();                              /* :299 (x10 variant) */
            if (!tract) {
                double ref_taps[8], amp = amplitude(cur[3]);
                memcpy(ref_taps, fricationTap, sizeof ref_taps);
                int ip = (int)cur[4];
                double complement = cur[4] - (double)ip, remainder = 1.0 - complement;
                for (int i = 0; i < 8; i++) {                /* TRMTubeModel.m:758-765 */
                    if (i == ip) {
                        fricationTap[i] = remainder * amp;
                        if ((i + 1) < 8) fricationTap[++i] = complement * amp;
                    } else
                        fricationTap[i] = 0.0;
                }
                for (int i = 0; i < 8; i++) {
                    double e = fabs(ref_taps[i] / 10.0 - fricationTap[i]);
                    if (e > tap_err) tap_err = e;
                }
            }
            calculateBandpassCoefficients();                 /* :300 */
            double lp_noise = noiseFilter(noise());          /* :305 */
            if (waveform == 0) updateWavetable(ax);          /* :308-309 */
            double pulse = oscillator(f0);                   /* :312 */
            double pulsed_noise = lp_noise * pulse;
            pulse = ax * ((pulse * (1.0 - breathinessFactor)) + (pulsed_noise * breathinessFactor));
            double sig;
            if (modulation) {
                double crossmix = ax * crossmixFactor;
                crossmix = (crossmix < 1.0) ? crossmix : 1.0;
                sig = (pulsed_noise * crossmix) + (lp_noise * (1.0 - crossmix));
            } else
                sig = lp_noise;
            sig = vocalTract(((pulse + (ah1 * sig)) * 0.125), bandpassFilter(sig)); /* :336-337 */
            sig += throat(pulse * 0.125);                    /* :341 (gain inside, tube.c:1716) */
            if (tract) sig = sig * 100;                      /* tube.c:1177 */
            tube[ntube++] = sig;
            dataFill(sig);                                   /* :346 */
            drain();
            for (int i = 0; i < 16; i++) cur[i] += delta[i]; /* :351 */
        }
    }
    if (nframes > 0) { flushBuffer(); drain(); }             /* :360 */

    FILE *fo = fopen(argv[2], "wb");
    if (!fo) return 2;
    int32_t i32; int64_t i64; uint32_t u32; double d;
    i32 = controlPeriod; fwrite(&i32, 4, 1, fo);
    i32 = sampleRate; fwrite(&i32, 4, 1, fo);
    i32 = padSize; fwrite(&i32, 4, 1, fo);
    i32 = numberTaps; fwrite(&i32, 4, 1, fo);
    u32 = timeRegisterIncrement; fwrite(&u32, 4, 1, fo);
    u32 = phaseIncrement; fwrite(&u32, 4, 1, fo);
    i64 = numberSamples; fwrite(&i64, 8, 1, fo);
    d = maximumSampleValue; fwrite(&d, 8, 1, fo);
    d = tap_err; fwrite(&d, 8, 1, fo);
    i64 = (int64_t)ntube; fwrite(&i64, 8, 1, fo);
    i64 = (int64_t)g_nout; fwrite(&i64, 8, 1, fo);
    fwrite(FIRCoef, sizeof(double), (size_t)numberTaps, fo);
    fwrite(h, sizeof(double), 3328, fo);
    fwrite(deltaH, sizeof(double), 3328, fo);
    fwrite(tube, sizeof(double), ntube, fo);
    fwrite(g_out, sizeof(float), g_nout, fo);
    fclose(fo);
    return 0;
}
