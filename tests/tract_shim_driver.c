/* Test driver for shim/tract_tube.c (TEST INFRASTRUCTURE): plays the role of TRAcT's Controller.m + CoreAudio callback
 * (Controller.m:73-100, 200, 231): starts the synthesizer, drains the circular buffer, changes parameters through
 * the pointers / setters the GUI uses, and dumps what it heard.
 *   usage: tract_shim_driver out.f32 nFirst nSecond [radius]     `radius`: the second part changes one radius only (the
 *   oscillator keeps its pitch, so the steady state can be compared index for index with tests/golden/tract_mode_ee_step) */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

int initializeSynthesizer(void);
float getCircBuff2(void);
void setRadius(float value, int index);
double *getGlotPitch(void);
double *getActualTubeLength(void);
int *getControlPeriod(void);
int *getSampleRate(void);
void shutdownSynthesizer(void);
extern int circBuff2Count;

int main(int argc, char **argv)
{
    if (argc < 4) return 64;
    long n1 = atol(argv[2]), n2 = atol(argv[3]);
    if (initializeSynthesizer()) return 2;
    float *buf = (float *)malloc((size_t)(n1 + n2) * sizeof(float));
    for (long i = 0; i < n1; i++) buf[i] = getCircBuff2();
    if (!(argc > 4 && !strcmp(argv[4], "radius"))) *getGlotPitch() = 7.0;      /* Controller.m:885 */
    setRadius(0.4f, 6);                          /* a slider of the tube view */
    for (long i = 0; i < n2; i++) buf[n1 + i] = getCircBuff2();
    FILE *f = fopen(argv[1], "wb");
    if (!f) return 3;
    fwrite(buf, sizeof(float), (size_t)(n1 + n2), f);
    fclose(f);
    printf("controlPeriod %d sampleRate %d actualTubeLength %.6f\n", *getControlPeriod(), *getSampleRate(), *getActualTubeLength());
    shutdownSynthesizer();
    return 0;
}
