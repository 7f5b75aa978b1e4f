/* Test driver for shim/tract_tube.c (TEST INFRASTRUCTURE): plays the role of TRAcT's Controller.m + CoreAudio callback
 * (Controller.m:73-100, 200, 231): starts the synthesizer, drains the circular buffer, changes parameters through
 * the pointers the GUI uses (Controller.m:231 writes *getGlotPitch()), and dumps what it heard.
 *
 *   usage: tract_shim_driver out.f32 total [R:name=value,name=value,...]...
 *
 * Every event "R:..." is a slider move at a KNOWN place: the driver first drains until it has heard R samples, then
 * waits until the circular buffer is full -- the synthesis thread is then blocked inside the control period that holds
 * output sample R + 8192 (tube.c:2414-2421 blocks the same way) and has read its parameters for that period -- and
 * writes the new values: they take effect with the next control period, whose number the test computes.
 * An event with R < 0 is applied before initializeSynthesizer() (the posture the program starts with).
 * names: glotPitch glotVol aspVol fricVol fricPos fricCF fricBW r0..r7 velum. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

int initializeSynthesizer(void);
float getCircBuff2(void);
double *getGlotPitch(void), *getGlotVol(void), *getAspVol(void), *getFricVol(void), *getFricPos(void), *getFricCF(void),
       *getFricBW(void), *getRadius(int), *getVelumRadius(void);
double *getActualTubeLength(void);
int *getControlPeriod(void);
int *getSampleRate(void);
int *getSliceSamples(void);
void shutdownSynthesizer(void);
extern int circBuff2Count;

static double *slot(const char *name)
{
    if (!strcmp(name, "glotPitch")) return getGlotPitch();
    if (!strcmp(name, "glotVol")) return getGlotVol();
    if (!strcmp(name, "aspVol")) return getAspVol();
    if (!strcmp(name, "fricVol")) return getFricVol();
    if (!strcmp(name, "fricPos")) return getFricPos();
    if (!strcmp(name, "fricCF")) return getFricCF();
    if (!strcmp(name, "fricBW")) return getFricBW();
    if (!strcmp(name, "velum")) return getVelumRadius();
    if (name[0] == 'r' && name[1] >= '0' && name[1] <= '7' && !name[2]) return getRadius(name[1] - '0');
    return NULL;
}

int main(int argc, char **argv)
{
    if (argc < 3) return 64;
    long total = atol(argv[2]), heard = 0;
    int started = 0;
    float *buf = (float *)malloc((size_t)total * sizeof(float));
    for (int a = 3; a < argc; a++) {
        char *ev = argv[a], *colon = strchr(ev, ':');
        if (!colon) return 64;
        long R = atol(ev);
        if (R >= 0) {
            if (!started && initializeSynthesizer()) return 2;
            started = 1;
            for (; heard < R && heard < total; heard++) buf[heard] = getCircBuff2();
            for (int spin = 0; *(volatile int *)&circBuff2Count != 8192; spin++) {     /* the thread fills the buffer and blocks */
                if (spin > 20000) { fprintf(stderr, "buffer never filled\n"); return 4; }
                usleep(500);
            }
        }
        for (char *tok = strtok(colon + 1, ","); tok; tok = strtok(NULL, ",")) {
            char *eq = strchr(tok, '=');
            if (!eq) return 64;
            *eq = 0;
            double *p = slot(tok);
            if (!p) { fprintf(stderr, "unknown parameter %s\n", tok); return 64; }
            *p = atof(eq + 1);
        }
    }
    if (!started && initializeSynthesizer()) return 2;
    for (; heard < total; heard++) buf[heard] = getCircBuff2();
    FILE *f = fopen(argv[1], "wb");
    if (!f) return 3;
    fwrite(buf, sizeof(float), (size_t)total, f);
    fclose(f);
    printf("controlPeriod %d sampleRate %d actualTubeLength %.6f slice %d\n", *getControlPeriod(), *getSampleRate(), *getActualTubeLength(),
           *getSliceSamples());
    shutdownSynthesizer();
    return 0;
}
