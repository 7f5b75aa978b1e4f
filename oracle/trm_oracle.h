/*
 * trm_oracle.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C, double-precision CPU restatement of GnuSpeech's Tube Resonance Model
 * (Frameworks/Tube).  It is the parity checker for the HIP path: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.  Nothing under
 * gnuspeech_amd/ links or calls it.
 *
 * Pinning: the reference ships no tests or golden outputs.  The
 * restatement is pinned against the reference's own compilable C tube
 * (Applications/TRAcT/tube.c) built by oracle/Makefile into oracle/_ref/ and driven by
 * oracle/ref_driver.c; the agreement is asserted by tests/test_oracle_vs_ref.py (when
 * /root/reference is present) and frozen into tests/golden/ by tests/golden/make_golden.py.
 */
#ifndef TRM_ORACLE_H
#define TRM_ORACLE_H

#include "../include/trm_c_api.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct trm_oracle_result {
    /* TRMSampleRateConverter output stream (TRMSampleRateConverter.m:206-214) */
    double  *samples;
    int32_t  numberSamples;
    double   maximumSampleValue;
    /* tube-rate signal handed to -dataFill: (TRMTubeModel.m:346), before conversion */
    double  *tubeSamples;
    int32_t  numberTubeSamples;
    trm_derived derived;
} trm_oracle_result;

/* -initWithInputData: + -synthesize (TRMTubeModel.m:186-260, 272-361).
 * frames = nframes x 16 doubles in file column order.  Returns a TRM_* code. */
int  trm_oracle_synthesize(const trm_input_params *params, const double *frames, size_t nframes,
                           int keep_tube_samples, trm_oracle_result *out);
/* The same in the loop order of Applications/TRAcT/tube.c (tube.c:1096-1190): frame f (f >= 1) is HELD for control period f,
 * frication taps x10 (tube.c:1371), tube-rate sample x100 before the converter (tube.c:1177).  Frame 0 is ignored, as
 * oracle/ref_driver.c `tract` ignores it. */
int  trm_oracle_synthesize_tract(const trm_input_params *params, const double *frames, size_t nframes,
                                 int keep_tube_samples, trm_oracle_result *out);
/* ... with every frame (from the second on) held for `slice` tube samples instead of a control period: tube.c's loop with
 * parameter changes on a grid of `slice` samples (what trm_stream_set_slice streams). */
int  trm_oracle_synthesize_tract_slices(const trm_input_params *params, const double *frames, size_t nframes, int32_t slice,
                                        int keep_tube_samples, trm_oracle_result *out);
void trm_oracle_result_free(trm_oracle_result *r);
/* `count` voices in a row on the calling thread (bench.py's cpu_baseline: one call per thread, no Python in the loop) */
int  trm_oracle_run_voices(const trm_input_params *params, const double *frames, size_t nframes, size_t nvoices,
                           size_t first, size_t count, uint64_t *samples_out);

/* derived constants only (TRMTubeModel.m:196-203, TRMSampleRateConverter.m:69-104) */
int  trm_oracle_derive(const trm_input_params *params, trm_derived *out);

/* oscillator FIR taps (TRMFIRFilter.m:37-98); returns tap count, writes <= cap taps */
int  trm_oracle_fir_taps(double beta, double gamma, double cutoff, double *taps, int cap);

/* sample-rate-converter tables h[3328], deltaH[3328] (TRMSampleRateConverter.m:110-131) */
void trm_oracle_src_tables(double *h, double *deltaH);

/* noise generator + one-zero low-pass (TRMUtility.m:71-85, TRMFilters.m:81-86):
 * lp[n] for n < count, seed reset as in TRMTubeModel.m:232-235 */
void trm_oracle_lp_noise(double *lp, size_t count);

/* dB -> amplitude, pitch -> Hz (TRMUtility.m:26-47) */
double trm_oracle_amplitude(double decibelLevel);
double trm_oracle_frequency(double pitch);

/* Output scaling (TRMTubeModel.m:370-389 file path, :515-533 WAV-data path):
 * interleaved int16 (host byte order) for `channels` channels. */
void trm_oracle_scale_int16(const trm_input_params *params, const double *samples, int32_t n,
                            double maximumSampleValue, int for_wav_data, int16_t *out);

/* -generateWAVData byte image (TRMTubeModel.m:509-593). Returns bytes written (0 if cap too small). */
size_t trm_oracle_wav_data(const trm_input_params *params, const double *samples, int32_t n,
                           double maximumSampleValue, uint8_t *buf, size_t cap);

/* .trm text parser (TRMDataList.m:43-247); *frames malloc'd (nframes x 16 doubles,
 * last row doubled). */
int  trm_oracle_parse_file(const char *path, trm_input_params *params, double **frames, size_t *nframes);


/* Control-track generation (oracle/evt_oracle.c): EventList.m:883-1061 + MMDriftGenerator.m. */
int trm_oracle_count_frames(const uint32_t *times, size_t n, const trm_intonation *s, size_t *nframes);
int trm_oracle_generate_frames(const uint32_t *times, const double *values, size_t n, const trm_intonation *s,
                               float *frames_out, size_t frames_cap, size_t *nframes);

#ifdef __cplusplus
}
#endif
#endif
