/* tract_tube.c -- TRAcT's synthesis back end (Applications/TRAcT/tube.c) over libtrm_hip.so: the functions of
 * Applications/TRAcT/tube.h that Controller.m and the CoreAudio callback use (tube.h:16-102), implemented on a
 * one-voice trm_stream (include/trm_c_api.h).  Drop-in for tube.c in the TRAcT target: remove tube.c, add this file,
 * link libtrm_hip.so.  Plain C + pthreads; it is compiled and exercised on the GPU box by tests/test_tract_shim.py.
 *
 *   tube.c                                              here
 *   ------------------------------------------------    ---------------------------------------------------------
 *   globals written through get*() pointers             the same globals (Controller.m:231 writes *getGlotPitch())
 *   initializeSynthesizer() (tube.c:580-680)            (re)creates the stream from the utterance-rate globals and
 *                                                       starts the synthesis thread once
 *   synthesize() thread (tube.c:1096-1190): one tube    synthesis thread: one slice (~1 ms) per trm_stream_push of
 *   sample per iteration from `current`, no             the current parameter set, in TRM_STREAM_MODE_TRACT: the set is
 *   interpolation, x10 frication taps (:1371), x100     HELD for the slice (a slider write steps at the next push, not
 *   (:1177), dataFill -> dataEmpty -> circBuff2         glides), frication taps x10, output x100; PCM -> circBuff2
 *   getCircBuff2() (tube.c:3197-3230), circBuff2Count   the same blocking pop and counter (Controller.m:88-91)
 *
 * What remains different from tube.c: a slider write takes effect at the next SLICE boundary (tube.c: at the next sample).
 * Round 4: the synthesis thread pushes one frame per slice of sampleRate / 1000 tube samples (trm_stream_set_slice; the
 * parameters are held, so the length of a "period" is free), i.e. a write is heard within a millisecond where whole
 * control periods made it up to 10 ms at TRAcT's 100 Hz.  The environment variable TRACT_SLICE_SAMPLES overrides the slice
 * (0 = whole control periods, as rounds 2-3 pushed; tests use it for moves at known periods).  The x100 gain is applied to the
 * converter's output (tube.c: to its input; the converter is linear); the converter output is the library's fp32.
 * tests/test_tract_shim.py plays Controller.m's part and checks what comes out of circBuff2 against the REFERENCE's
 * tube.c run in its own loop order (tests/golden/tract_mode_*.npz) over whole utterances, slider moves included.
 */
#include <math.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../include/trm_c_api.h"

#define TOTAL_REGIONS 8
#define TOTAL_NASAL_SECTIONS 6
#define CIRC_BUFF2_SIZE 8192

/* ---- posture-rate parameters (tube.c:290-314) and their defaults: "ee" */
static double glotPitch = -0.0, glotVol = 60.0, aspVol = 0.0, fricVol = 0.0, fricPos = 8.0, fricCF = 5000.0, fricBW = 250.0;
static double radius[TOTAL_REGIONS] = {0.8, 1.67, 1.905, 1.985, 0.81, 0.495, 0.73, 1.485};
static double velum = 0.0;
static double glotPitchDef = 0.0, glotVolDef = 60.0, aspVolDef = 0.0, fricVolDef = 0.0, fricPosDef = 8.0, fricCFDef = 5000.0,
              fricBWDef = 250.0, velumDef = 0.0;
static double radiusDef[TOTAL_REGIONS] = {0.8, 1.67, 1.905, 1.985, 0.81, 0.495, 0.73, 1.485};

/* ---- utterance-rate parameters (tube.c:326-352) */
static double apScale = 2.5, balance = 0, breathiness = 2.5, length = 17, lossFactor = 0.8, mixOffset = 48.0, mouthCoef = 4000.0,
              noseCoef = 4000.0, temperature = 32, throatCutoff = 1500.0, throatVol = 6.0, tnMax = 40, tnMin = 16, tp = 35, volume = 60;
static double noseRadius[TOTAL_NASAL_SECTIONS] = {1.35, 1.35, 1.7, 1.7, 1.3, 0.9};
static int modulation = 1, waveform = 0;
static float controlRate = 100, outputRate = 44100;
static double apScaleDef = 2.5, balanceDef = 0, breathinessDef = 2.5, lengthDef = 17, lossFactorDef = 0.8, mixOffsetDef = 48.0,
              mouthCoefDef = 4000.0, noseCoefDef = 4000.0, temperatureDef = 32, throatCutoffDef = 1500.0, throatVolDef = 6.0,
              tnMaxDef = 40, tnMinDef = 16, tpDef = 35, volumeDef = 60;
static double noseRadiusDef[TOTAL_NASAL_SECTIONS] = {1.35, 1.35, 1.7, 1.7, 1.3, 0.9};
static int modulationDef = 1, waveformDef = 0;

/* ---- derived (tube.c:596-612) */
static double actualTubeLength;
static int controlPeriod, sampleRate;
static int sliceSamples;                       /* tube samples per push (0: a control period) */
static double wavetable[512];

/* ---- the circular buffer between the synthesis thread and the audio callback (tube.c:3141-3230) */
static float circBuff2[CIRC_BUFF2_SIZE];
static size_t cbIn, cbOut;
int circBuff2Count = 0;
static pthread_mutex_t cbMutex = PTHREAD_MUTEX_INITIALIZER;
static pthread_cond_t cbCond = PTHREAD_COND_INITIALIZER;

static trm_stream *stream;
static pthread_mutex_t streamMutex = PTHREAD_MUTEX_INITIALIZER;
static int threadFlag = 0;
static volatile int stopFlag = 0;

/* setters (tube.h:16-40) */
void setGlotPitch(float v) { glotPitch = v; }
void setGlotVol(float v) { glotVol = v; }
void setAspVol(float v) { aspVol = v; }
void setFricVol(float v) { fricVol = v; }
void setfricPos(float v) { fricPos = v; }
void setFricCF(float v) { fricCF = v; }
void setFricBW(float v) { fricBW = v; }
void setRadius(float v, int i) { radius[i] = v; }
void setVelum(float v) { velum = v; }
void setVolume(double v) { volume = v; }
void setWaveformType(int v) { waveform = v; }
void setTp(double v) { tp = v; }
void setTnMin(double v) { tnMin = v; }
void setTnMax(double v) { tnMax = v; }
void setBreathiness(double v) { breathiness = v; }
void setLength(double v) { length = v; }
void setTemperature(double v) { temperature = v; }
void setLossFactor(double v) { lossFactor = v; }
void setApScale(double v) { apScale = v; }
void setMouthCoef(double v) { mouthCoef = v; }
void setNoseCoef(double v) { noseCoef = v; }
void setNoseRadius(double v, int i) { noseRadius[i] = v; }
void setThroatCoef(double v) { throatCutoff = v; }
void setModulation(int v) { modulation = v; }
void setMixOffset(double v) { mixOffset = v; }

/* getters: pointers to the live values and to the defaults (tube.h:43-100) */
#define GETTER(fn, var) double *fn(void) { return &var; }
GETTER(getGlotPitch, glotPitch) GETTER(getGlotVol, glotVol) GETTER(getAspVol, aspVol) GETTER(getFricVol, fricVol)
GETTER(getFricPos, fricPos) GETTER(getFricCF, fricCF) GETTER(getFricBW, fricBW) GETTER(getVelumRadius, velum)
GETTER(getVolume, volume) GETTER(getBalance, balance) GETTER(getTp, tp) GETTER(getTnMin, tnMin) GETTER(getTnMax, tnMax)
GETTER(getBreathiness, breathiness) GETTER(getLength, length) GETTER(getTemperature, temperature)
GETTER(getLossFactor, lossFactor) GETTER(getApScale, apScale) GETTER(getMouthCoef, mouthCoef) GETTER(getNoseCoef, noseCoef)
GETTER(getThroatCutoff, throatCutoff) GETTER(getThroatVol, throatVol) GETTER(getMixOffset, mixOffset)
GETTER(getActualTubeLength, actualTubeLength)
GETTER(getGlotPitchDefault, glotPitchDef) GETTER(getGlotVolDefault, glotVolDef) GETTER(getAspVolDefault, aspVolDef)
GETTER(getFricVolDefault, fricVolDef) GETTER(getFricPosDefault, fricPosDef) GETTER(getFricCFDefault, fricCFDef)
GETTER(getFricBWDefault, fricBWDef) GETTER(getVelumRadiusDefault, velumDef) GETTER(getVolumeDefault, volumeDef)
GETTER(getBalanceDefault, balanceDef) GETTER(getTpDefault, tpDef) GETTER(getTnMinDefault, tnMinDef)
GETTER(getTnMaxDefault, tnMaxDef) GETTER(getBreathinessDefault, breathinessDef) GETTER(getLengthDefault, lengthDef)
GETTER(getTemperatureDefault, temperatureDef) GETTER(getLossFactorDefault, lossFactorDef) GETTER(getApScaleDefault, apScaleDef)
GETTER(getMouthCoefDefault, mouthCoefDef) GETTER(getNoseCoefDefault, noseCoefDef) GETTER(getThroatCutoffDefault, throatCutoffDef)
GETTER(getThroatVolDefault, throatVolDef) GETTER(getMixOffsetDefault, mixOffsetDef)
double *getRadius(int i) { return &radius[i]; }
double *getRadiusDefault(int i) { return &radiusDef[i]; }
double *getNoseRadius(int i) { return &noseRadius[i]; }
double *getNoseRadiusDefault(int i) { return &noseRadiusDef[i]; }
int *getWaveform(void) { return &waveform; }
int *getWaveformDefault(void) { return &waveformDef; }
int *getModulation(void) { return &modulation; }
int *getModulationDefault(void) { return &modulationDef; }
int *getControlPeriod(void) { return &controlPeriod; }
int *getSliceSamples(void) { return &sliceSamples; }          /* (no tube.h counterpart: what a push stands for) */
float *getControlRate(void) { return &controlRate; }
int *getSampleRate(void) { return &sampleRate; }
double *getWavetable(int i) { return &wavetable[i & 511]; }
int *getThreadFlag(void) { return &threadFlag; }

static void put_sample(float x)      /* blocks while the buffer is full (tube.c:2414-2421) */
{
    pthread_mutex_lock(&cbMutex);
    while (circBuff2Count == CIRC_BUFF2_SIZE && !stopFlag) pthread_cond_wait(&cbCond, &cbMutex);
    circBuff2[cbIn] = x;
    cbIn = (cbIn + 1) % CIRC_BUFF2_SIZE;
    circBuff2Count++;
    pthread_mutex_unlock(&cbMutex);
    pthread_cond_signal(&cbCond);
}

float getCircBuff2(void)            /* blocks while the buffer is empty (tube.c:3197-3230) */
{
    pthread_mutex_lock(&cbMutex);
    while (circBuff2Count == 0) pthread_cond_wait(&cbCond, &cbMutex);
    float x = circBuff2[cbOut];
    cbOut = (cbOut + 1) % CIRC_BUFF2_SIZE;
    circBuff2Count--;
    pthread_mutex_unlock(&cbMutex);
    pthread_cond_signal(&cbCond);
    return x;
}

static void *synthesize(void *unused)          /* tube.c:1096-1190 */
{
    (void)unused;
    float *out = NULL;
    size_t cap = 0;
    for (;;) {
        if (stopFlag) break;
        pthread_mutex_lock(&streamMutex);
        float f[16] = {(float)glotPitch, (float)glotVol, (float)aspVol, (float)fricVol, (float)fricPos, (float)fricCF, (float)fricBW};
        for (int i = 0; i < TOTAL_REGIONS; i++) f[7 + i] = (float)radius[i];
        f[15] = (float)velum;
        size_t n = trm_stream_samples_for_push(stream, 1);
        if (n > cap) { out = (float *)realloc(out, n * sizeof(float)); cap = n; }
        uint32_t got = 0;
        int rc = trm_stream_push(stream, f, 1, out, cap ? cap : 1, &got, NULL);
        pthread_mutex_unlock(&streamMutex);
        if (rc) { fprintf(stderr, "tract_tube: %s\n", trm_last_error()); break; }
        for (uint32_t i = 0; i < got; i++) put_sample(out[i]);                 /* (x100, tube.c:1177: done by the mode) */
    }
    free(out);
    return NULL;
}

int initializeSynthesizer(void)                 /* tube.c:580-680 */
{
    trm_input_params p;
    memset(&p, 0, sizeof p);
    p.outputFileFormat = 1;
    p.outputRate = outputRate;
    p.controlRate = controlRate;
    p.volume = volume; p.channels = 2; p.balance = balance;
    p.waveform = waveform; p.tp = tp; p.tnMin = tnMin; p.tnMax = tnMax; p.breathiness = breathiness;
    p.length = length; p.temperature = temperature; p.lossFactor = lossFactor; p.apScale = apScale;
    p.mouthCoef = mouthCoef; p.noseCoef = noseCoef;
    for (int i = 0; i < TOTAL_NASAL_SECTIONS; i++) p.noseRadius[i] = noseRadius[i];
    p.throatCutoff = throatCutoff; p.throatVol = throatVol; p.usesModulation = modulation; p.mixOffset = mixOffset;
    trm_derived d;
    if (trm_derive(&p, &d)) { fprintf(stderr, "tube.c:538Illegal tube length.\n"); return -1; }     /* tube.c:613-616 */
    controlPeriod = d.controlPeriod;
    sampleRate = d.sampleRate;
    actualTubeLength = d.actualTubeLength;
    for (int i = 0; i < 512; i++) {             /* the display copy of the glottal pulse (tube.c:701-760) */
        int div1 = (int)rint(512 * (tp / 100.0)), div2 = (int)rint(512 * ((tp + tnMax) / 100.0));
        double x;
        if (i < div1) { x = (double)i / div1; wavetable[i] = (3 * x * x) - (2 * x * x * x); }
        else if (i < div2) { x = (double)(i - div1) / (div2 - div1); wavetable[i] = 1.0 - x * x; }
        else wavetable[i] = 0.0;
    }
    pthread_mutex_lock(&streamMutex);
    trm_stream *ns = NULL;
    int rc = trm_stream_create(&p, -1, 1, &ns);
    if (rc == 0) rc = trm_stream_set_mode(ns, TRM_STREAM_MODE_TRACT);          /* tube.c's own loop order */
    {
        /* tube.c reads `current` every sample (tube.c:1121-1136): push the parameter set every millisecond, not every period */
        const char *e = getenv("TRACT_SLICE_SAMPLES");
        sliceSamples = e ? atoi(e) : (int)lrint(sampleRate / 1000.0);
        if (sliceSamples != 0 && sliceSamples < 4) sliceSamples = 4;
        if (rc == 0) rc = trm_stream_set_slice(ns, (uint32_t)sliceSamples);
    }
    if (rc == 0) {
        if (stream) trm_stream_destroy(stream);
        stream = ns;
    } else if (ns)
        trm_stream_destroy(ns);
    pthread_mutex_unlock(&streamMutex);
    if (rc) { fprintf(stderr, "tract_tube: %s\n", trm_last_error()); return -1; }
    if (threadFlag == 0) {                      /* tube.c:660-670 */
        pthread_t tid;
        threadFlag = 1;
        if (pthread_create(&tid, NULL, synthesize, NULL) != 0) return -1;
        pthread_detach(tid);
    }
    return 0;
}

void shutdownSynthesizer(void)                  /* (no counterpart: tube.c's thread runs until the process ends) */
{
    stopFlag = 1;
    pthread_cond_broadcast(&cbCond);
}
