/*
 * trm_c_api.h -- C ABI of libtrm_hip.so, the MI355X (gfx950) Tube Resonance Model.
 *
 * This header is the drop-in boundary for ONE path of grrrr/GnuSpeech: the per-sample
 * Tube Resonance Model loop `-[TRMTubeModel synthesize]` (Frameworks/Tube) and the
 * data model / output writers on either side of it.  Every entry point names the
 * reference interface it replaces (file:line relative to the GnuSpeech tree).  The
 * reference surface is Objective-C; the ABI below is what an Objective-C shim (see
 * INTEGRATION.md and shim/) binds: plain pointers and sizes, no C++/torch types.
 *
 * Threading: handles are independent; calls on one handle are blocking and must not
 * overlap (same contract as a TRMTubeModel instance, TRMTubeModel.m:133-184).
 */
#ifndef TRM_C_API_H
#define TRM_C_API_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Frameworks/Tube/TRMTubeModel.h:6-25 */
#define TRM_TOTAL_REGIONS        8
#define TRM_TOTAL_NASAL_SECTIONS 6
#define TRM_FRAME_VALUES         16   /* TRMParameters.h:9-17: 7 scalars + radius[8] + velum */

/* Frameworks/Tube/TRMInputParameters.h:7-20 */
enum { TRM_SOUND_FILE_FORMAT_AU = 0, TRM_SOUND_FILE_FORMAT_AIFF = 1, TRM_SOUND_FILE_FORMAT_WAVE = 2 };
enum { TRM_WAVEFORM_PULSE = 0, TRM_WAVEFORM_SINE = 1 };

/* error codes (the reference returns nil/NO and prints to stderr; TRMTubeModel.m:204-207) */
enum {
    TRM_OK             = 0,
    TRM_EINVAL         = 1,   /* NULL / malformed argument                                  */
    TRM_EINVAL_LENGTH  = 2,   /* tube length <= 0           (TRMTubeModel.m:204-207 -> nil)   */
    TRM_EFIR           = 3,   /* FIR design failure         (TRMFIRFilter.m:49-51 -> nil)     */
    TRM_ENOMEM         = 4,
    TRM_EHIP           = 5,   /* HIP runtime error; text via trm_last_error()               */
    TRM_ENODEVICE      = 6,   /* no gfx950 device visible: the product path never falls back */
    TRM_EIO            = 7,   /* file open / read / write   (TRMDataList.m:45-49 -> NO)       */
    TRM_EPARSE         = 8,   /* truncated utterance-rate header (TRMDataList.m:53-214)      */
    TRM_ESILENT        = 9,   /* maximumSampleValue == 0    (TRMTubeModel.m:511 assert)       */
    TRM_ERANGE         = 10   /* rates or glottal-pulse shape (tp + tnMax > 100 %) outside what is supported */
};

/* TRMInputParameters (Frameworks/Tube/TRMInputParameters.h:26-54): the 26 utterance-rate
 * fields, same names, same types (outputRate/controlRate are float in the reference). */
typedef struct trm_input_params {
    int32_t outputFileFormat;     /* 0=AU 1=AIFF 2=WAVE                         */
    float   outputRate;           /* 22050 / 44100                              */
    float   controlRate;          /* 1-1000 input tables / s                    */
    double  volume;               /* master volume 0-60 dB                      */
    int32_t channels;             /* 1 or 2                                     */
    double  balance;              /* -1..+1                                     */
    int32_t waveform;             /* 0=pulse 1=sine                             */
    double  tp;                   /* % glottal pulse rise time                  */
    double  tnMin;                /* % fall time minimum                        */
    double  tnMax;                /* % fall time maximum                        */
    double  breathiness;          /* % glottal source breathiness               */
    double  length;               /* nominal tube length cm                     */
    double  temperature;          /* deg C                                      */
    double  lossFactor;           /* junction loss %                            */
    double  apScale;              /* aperture scaling radius cm                 */
    double  mouthCoef;            /* mouth aperture coefficient (Hz)            */
    double  noseCoef;             /* nose aperture coefficient (Hz)             */
    double  noseRadius[TRM_TOTAL_NASAL_SECTIONS]; /* [0] unused (TRMDataList.m:178) */
    double  throatCutoff;         /* Hz                                         */
    double  throatVol;            /* dB                                         */
    int32_t usesModulation;       /* pulse modulation of noise                  */
    double  mixOffset;            /* noise crossmix offset dB                   */
} trm_input_params;

/* TRMParameters (Frameworks/Tube/TRMParameters.h:9-17): one control-rate frame,
 * 16 doubles in .trm file column order (TRMDataList.m:223-233). */
typedef struct trm_parameters {
    double glottalPitch;
    double glottalVolume;
    double aspirationVolume;
    double fricationVolume;
    double fricationPosition;
    double fricationCenterFrequency;
    double fricationBandwidth;
    double radius[TRM_TOTAL_REGIONS];
    double velum;
} trm_parameters;

/* Values TRMTubeModel derives in -initWithInputData: (TRMTubeModel.m:196-241) and
 * TRMSampleRateConverter -initWithInputRate:outputRate: (TRMSampleRateConverter.m:69-104);
 * what -printInputData prints (TRMTubeModel.m:595-605). */
typedef struct trm_derived {
    int32_t  controlPeriod;
    int32_t  sampleRate;            /* tube rate */
    double   actualTubeLength;
    double   sampleRateRatio;
    uint32_t timeRegisterIncrement;
    uint32_t phaseIncrement;        /* down-sampling only */
    int32_t  padSize;
    int32_t  firTaps;               /* oscillator FIR taps (49 for the shipped beta/gamma/cutoff) */
} trm_derived;

const char *trm_strerror(int code);
const char *trm_last_error(void);          /* thread-local detail text of the last failure */

/* ------------------------------------------------------------------------------------
 * Data model + text format: TRMDataList (Frameworks/Tube/TRMDataList.m:32-40, 43-247).
 * 26 header lines (first token of each), then rows of 16 values; the file path doubles
 * the last row (TRMDataList.m:239-241).  *frames is malloc'd; release with trm_free().
 * ------------------------------------------------------------------------------------ */
int  trm_data_list_read_file(const char *path, trm_input_params *params,
                             trm_parameters **frames, size_t *nframes);
/* Writer of the same format: MMSynthesisParameters -parameterString
 * (MonetModel/MMSynthesisParameters.m:278-310) + TRMParameters -valuesString
 * (TRMParameters.m:26-43); what Monet dumps to /tmp/Monet.parameters. */
int  trm_data_list_write_file(const char *path, const trm_input_params *params,
                              const trm_parameters *frames, size_t nframes);
void trm_free(void *p);

/* ------------------------------------------------------------------------------------
 * TRMTubeModel (Frameworks/Tube/TRMTubeModel.h:29-40): one tube per utterance.
 * ------------------------------------------------------------------------------------ */
typedef struct trm_tube trm_tube;

/* -initWithInputData: (TRMTubeModel.m:186-260).  device < 0 => current HIP device. */
int  trm_tube_create(const trm_input_params *params, int device, trm_tube **tube);
void trm_tube_destroy(trm_tube *tube);
int  trm_tube_derived(const trm_tube *tube, trm_derived *out);

/* -printInputData (TRMTubeModel.m:595-605): -[TRMDataList printInputParameters] (TRMDataList.m:251-292), the three derived
 * values, -[TRMDataList printControlRateInputTable] (TRMDataList.m:294-330), to stdout in the reference's formats. */
int  trm_tube_print_input_data(const trm_tube *tube, const trm_parameters *frames, size_t nframes);

/* -synthesize (TRMTubeModel.m:272-361) over inputData.values = frames[0..nframes):
 * N frames -> N-1 control periods; 0 frames is a silent no-op (:274-277). */
int  trm_tube_synthesize(trm_tube *tube, const trm_parameters *frames, size_t nframes);

/* TRMSampleRateConverter numberSamples / maximumSampleValue / resampledData
 * (TRMSampleRateConverter.m:206-214,312-315).  The pointer stays valid until the next
 * synthesize or destroy. */
size_t       trm_tube_number_samples(const trm_tube *tube);
double       trm_tube_maximum_sample_value(const trm_tube *tube);
const float *trm_tube_samples(const trm_tube *tube);

/* -saveOutputToFile:error: (TRMTubeModel.m:365-490): scale, balance, int16, AU/AIFF/WAVE
 * container chosen by params.outputFileFormat. */
int  trm_tube_save_output_to_file(trm_tube *tube, const char *filename);
/* -generateWAVData (TRMTubeModel.m:509-593).  Call with buf==NULL to get the size. */
int  trm_tube_generate_wav_data(trm_tube *tube, uint8_t *buf, size_t cap, size_t *len);

/* ------------------------------------------------------------------------------------
 * Batch entry (no reference equivalent: the reference builds one tube per utterance,
 * TRMSynthesizer.m:118-136; GnuTTSServer calls it once per utterance).  One call runs
 * V independent tubes that share one trm_input_params.
 * ------------------------------------------------------------------------------------ */
typedef struct trm_batch trm_batch;

int  trm_batch_create(const trm_input_params *params, int device, trm_batch **batch);
void trm_batch_destroy(trm_batch *batch);
int  trm_batch_derived(const trm_batch *batch, trm_derived *out);

/* Output samples a voice of `nframes` frames produces (exact, integer-only). */
size_t trm_batch_samples_for_frames(const trm_batch *batch, size_t nframes);
/* Same, and the derived constants, without a device (sizing / sharding on hosts that only plan). */
int    trm_derive(const trm_input_params *params, trm_derived *out);
size_t trm_samples_for_frames(const trm_input_params *params, size_t nframes);

/* Host-buffer form: frames = concatenated rows, voice v owns rows
 * [frame_offset[v], frame_offset[v]+nframes[v]); out receives voice v's fp32 PCM at
 * out + out_offset[v] (caller sizes it with trm_batch_samples_for_frames);
 * number_samples[v] / max_sample[v] are the converter's numberSamples and
 * maximumSampleValue.  Includes H2D/D2H. */
int  trm_batch_synthesize_host(trm_batch *batch, size_t nvoices,
                               const float *frames, const uint64_t *frame_offset,
                               const uint32_t *nframes,
                               float *out, const uint64_t *out_offset,
                               uint32_t *number_samples, float *max_sample);

/* The same, returning what -saveOutputToFile: / -generateWAVData put into their containers (TRMTubeModel.m:370-389,
 * 515-540): int16 PCM scaled per voice by 32767/max * amplitude(volume), mono or -- params->channels == 2 --
 * interleaved stereo with the balance applied (for_wav_data != 0: -generateWAVData's variant without the x2).  Voice v's
 * first value is out16[out_offset[v] * channels]; half the bytes of the fp32 form cross PCIe. */
int  trm_batch_synthesize_host_int16(trm_batch *batch, size_t nvoices,
                                     const float *frames, const uint64_t *frame_offset,
                                     const uint32_t *nframes,
                                     int16_t *out16, const uint64_t *out_offset,
                                     uint32_t *number_samples, float *max_sample, int for_wav_data);

/* Device-buffer form (all pointers are HIP device pointers on the batch's device;
 * stream is a hipStream_t or NULL).  Asynchronous on `stream`.  max_nframes = the largest d_nframes[v]: the
 * voice-independent tables are sized from it, and a voice that claims more frames is cut to it. */
int  trm_batch_synthesize_device(trm_batch *batch, size_t nvoices,
                                 const float *d_frames, const uint64_t *d_frame_offset,
                                 const uint32_t *d_nframes, uint32_t max_nframes,
                                 float *d_out, const uint64_t *d_out_offset,
                                 uint32_t *d_number_samples, float *d_max_sample,
                                 void *stream);

/* ---------------------------------------------------------------------------------------------
 * Several GPUs from one process (voices are independent units -- contiguous voice ranges per device,
 * private buffers, one host thread and one stream per device, no collective).  bench.py uses one PROCESS per GPU
 * instead; the shard boundaries are the same function.
 * --------------------------------------------------------------------------------------------- */
/* bounds[0..nshards]: shard g = voices [bounds[g], bounds[g+1]), contiguous, covering all voices, balanced by frame
 * count (a voice's cost is proportional to its frames).  Host-only. */
int  trm_shard_voices(const uint32_t *nframes, size_t nvoices, size_t nshards, size_t *bounds);

typedef struct trm_multi trm_multi;
/* One trm_batch per entry of `devices` (a device may be listed more than once: its shards then share it). */
int  trm_multi_create(const trm_input_params *params, const int *devices, size_t ndevices, trm_multi **out);
void trm_multi_destroy(trm_multi *m);
/* trm_batch_synthesize_host over all devices: same arguments and results.  Each device receives only its shard's
 * frames and returns only its shard's PCM, so the output ranges of different shards must not interleave (voices laid
 * out in index order, as TRMBatch does, satisfy this); TRM_EINVAL otherwise. */
int  trm_multi_synthesize_host(trm_multi *m, size_t nvoices, const float *frames, const uint64_t *frame_offset,
                               const uint32_t *nframes, float *out, const uint64_t *out_offset,
                               uint32_t *number_samples, float *max_sample);

/* trm_batch_synthesize_host_int16 over all devices. */
int  trm_multi_synthesize_host_int16(trm_multi *m, size_t nvoices, const float *frames, const uint64_t *frame_offset,
                                     const uint32_t *nframes, int16_t *out16, const uint64_t *out_offset,
                                     uint32_t *number_samples, float *max_sample, int for_wav_data);

/* Output normalisation on device, TRMTubeModel.m:370-389,420-484: int16 mono/stereo
 * from fp32 PCM with per-voice scale = 32767/max * amplitude(volume). */
int  trm_batch_scale_to_int16_device(trm_batch *batch, size_t nvoices,
                                     const float *d_pcm, const uint64_t *d_out_offset,
                                     const uint32_t *d_number_samples, const float *d_max_sample,
                                     int16_t *d_int16, int for_wav_data, void *stream);

/* Sound files composed on the device: for every voice the complete file image -- the container's header
 * (params->outputFileFormat: AU 24 bytes, AIFF 54, WAVE 44) followed by the int16 payload in the container's byte order, scaled
 * as -saveOutputToFile:error: scales it (TRMTubeModel.m:370-389) -- at d_files + d_file_offset[v] (bytes).  An image is
 * trm_sound_file_size(params, numberSamples) bytes: byte for byte what trm_write_sound_file writes for the same samples.  One
 * copy (or a write() from a mapped buffer) then gives ready files; nothing but finished containers crosses PCIe. */
size_t trm_sound_file_size(const trm_input_params *params, size_t nsamples);
int  trm_batch_sound_files_device(trm_batch *batch, size_t nvoices, const float *d_pcm, const uint64_t *d_out_offset,
                                  const uint32_t *d_number_samples, const float *d_max_sample, uint8_t *d_files,
                                  const uint64_t *d_file_offset, void *stream);

/* -saveOutputToFile:error: (TRMTubeModel.m:365-490) for one voice of a batch: writes `n` fp32 samples
 * with their maximumSampleValue as the AU / AIFF / WAVE file params->outputFileFormat names (int16, the
 * reference's scale, balance and byte order).  Host-side container code; the samples come from
 * trm_batch_synthesize_*. */
int  trm_write_sound_file(const trm_input_params *params, const float *samples, size_t n, float maximumSampleValue,
                          const char *filename);

/* ---------------------------------------------------------------------------------------------
 * Control-track generation at 250 Hz: the step in front of the tube.
 * Replaces -[EventList generateOutputInTimeRange:forSynthesizer:parameterLogger:]
 * (Frameworks/GnuSpeech/MonetModel/EventList.m:883-1061) with MMDriftGenerator -generateDrift
 * (MMDriftGenerator.m:65-78): piece-wise linear interpolation of an utterance's event list
 * (events = time in ms + 36 values, NaN = "no target here") into the 16-column frames the tube
 * consumes, one frame every 4 ms.  On the device the frames land directly in the tube's frame
 * buffer, so a batch goes from event lists to PCM without the frames ever crossing PCIe.
 * Values 0..15 are the tube parameters, 16..31 their special-event offsets, 32 the intonation
 * contour (semitones), 33..35 the smooth-intonation slopes (EventList.m:931-959, 1045-1053). */
#define TRM_EVENT_VALUES 36

typedef struct trm_intonation {
    int32_t useMicroIntonation;    /* MMIntonation.h:11; off -> table[0] starts from 0 (EventList.m:974-975) */
    int32_t useMacroIntonation;    /* :10; adds the contour, value 32                      (EventList.m:978-981) */
    int32_t useSmoothIntonation;   /* :12; contour advanced by the cubic slopes 33..35     (EventList.m:1012-1015) */
    int32_t useDrift;              /* :14; adds MMDriftGenerator's low-passed noise        (EventList.m:976-977) */
    float   driftDeviation;        /* :15 semitones */
    float   driftCutoff;           /* :16 Hz */
    double  pitchMean;             /* MMSynthesisParameters.h:34 `pitch`, added last       (EventList.m:983) */
    uint32_t timeQuantization;     /* ms; the drift generator's rate is 1000 / this        (EventList.m:903) */
    uint32_t startTime_ms;         /* the time range: frames are emitted for start <= t <= end; */
    uint32_t endTime_ms;           /*   end == 0 and start == 0 means "everything"         (EventList.m:892-899) */
    float   driftSeed;             /* the drift generator's seed at the start of this utterance; 0 = MMDriftGenerator's
                                    * initial 0.7892347 (a fresh EventList).  The reference keeps ONE generator per EventList
                                    * and only -init sets its seed ("And seed is not changed...", MMDriftGenerator.m:41-58):
                                    * the second utterance of a list continues the sequence.  A caller reproduces that by
                                    * passing trm_drift_seed_after(seed, frames generated) of the utterance before. */
} trm_intonation;

/* The drift generator's seed after `ngenerated` calls of -generateDrift from `seed` (0 = the initial seed): one call per
 * 4 ms step of -generateOutputInTimeRange: whatever the time range, i.e. trm_events_count_frames() with start = end = 0
 * (MMDriftGenerator.m:65-78, EventList.m:970-977). */
float trm_drift_seed_after(float seed, size_t ngenerated);

/* Number of frames the generator emits for an event list with these event times (exact: follows the
 * loop's time stepping, EventList.m:979-1027).  nevents < 2 -> 0 (the reference indexes event 1). */
int  trm_events_count_frames(const uint32_t *event_times, size_t nevents, const trm_intonation *settings,
                             size_t *nframes);

/* Device entry.  Layout in HBM:
 *   d_event_times   u32 [sum nevents]        voice v owns entries event_offset[v] .. +nevents[v]
 *   d_event_values  f64 [sum nevents][36]    same indexing, NaN = absent
 *   d_frames        f32 [sum nframes][16]    voice v writes rows frame_offset[v] .. +nframes[v] where
 *                                            nframes[v] = trm_events_count_frames(...) (the caller sizes it)
 * One settings struct for the whole batch.  d_nframes_out[v] receives the number of rows written. */
int  trm_batch_generate_frames_device(trm_batch *batch, size_t nvoices,
                                      const uint32_t *d_event_times, const double *d_event_values,
                                      const uint64_t *d_event_offset, const uint32_t *d_nevents,
                                      const trm_intonation *settings,
                                      float *d_frames, const uint64_t *d_frame_offset, uint32_t *d_nframes_out,
                                      void *stream);

/* Host-buffer form of one utterance (H2D + kernel + D2H): frames_out has room for frames_cap rows. */
int  trm_batch_generate_frames_host(trm_batch *batch, const uint32_t *event_times, const double *event_values,
                                    size_t nevents, const trm_intonation *settings,
                                    float *frames_out, size_t frames_cap, size_t *nframes);

/* ---------------------------------------------------------------------------------------------
 * Streaming synthesis: an utterance delivered in chunks of control frames, PCM returned
 * per chunk, with the tube, oscillator, filter and converter state carried on the device from one chunk to
 * the next.  How the utterance is cut into chunks does not matter, bit for bit, and the streamed utterance
 * equals what trm_batch_synthesize_* returns for it at once to rounding (same sample count; both tested), so a
 * server can start playing after the first chunk.
 *
 * This is also what TRAcT's real-time loop needs (Applications/TRAcT/tube.c:1096-1190: a thread that keeps
 * synthesizing from the `current` parameter set into a circular buffer the CoreAudio callback drains,
 * tube.c:2348-2420, Controller.m:73-100): shim/tract_tube.c implements tube.h's setters/getters over a
 * one-voice stream and pushes the current parameters every control period.
 *
 * All voices of a stream advance together (same number of frames per push).  Output rates above the tube rate
 * (44.1 / 22.05 kHz for the shipped voices) and below it (16 / 8 kHz: the down-sampling branch) both stream.  Streams of
 * fewer voices than fill the chip (8192 on MI355X) run the four-lane kernel form, larger ones -- and those whose
 * converter makes more than four outputs per tube sample (96 kHz output) -- the one-voice-per-lane form; the form is
 * fixed when the stream is created (trm_stream_kernel). */
typedef struct trm_stream trm_stream;
int  trm_stream_create(const trm_input_params *params, int device, size_t nvoices, trm_stream **out);
void trm_stream_destroy(trm_stream *stream);
/* Whose sample loop the stream follows.  Applications/TRAcT/tube.c's real-time loop (tube.c:1096-1190) is the ancestor of
 * Frameworks/Tube's and differs from it in three documented ways; TRM_STREAM_MODE_TRACT reproduces them so that
 * shim/tract_tube.c sounds like tube.c:
 *   - no control-rate interpolation: tube.c converts `current.*` every sample (tube.c:1121-1136), a slider write takes effect at
 *     once.  Here EVERY pushed frame (the first one too) is one control period of HELD parameters: a change steps at
 *     the push boundary (TRM_STREAM_MODE_FRAMEWORK: the first frame is the starting point and every later frame one period
 *     interpolated from the frame before it, TRMTubeModel.m:611-688);
 *   - the frication taps carry ten times the amplitude (tube.c:1371 vs TRMTubeModel.m:750);
 *   - the output is 100 times louder (tube.c:1177; applied to the converter's output here, before it there: linear).
 * Only between utterances (before the first push or after trm_stream_finish); TRM_EINVAL otherwise. */
enum { TRM_STREAM_MODE_FRAMEWORK = 0, TRM_STREAM_MODE_TRACT = 1 };
int  trm_stream_set_mode(trm_stream *stream, int mode);
/* TRM_STREAM_MODE_TRACT only: the tube samples one pushed frame stands for (default: a control period).  tube.c reads its
 * parameter set every SAMPLE (tube.c:1121-1136), so a slider write is heard at once; with whole control periods per push it is
 * heard at the next period boundary, up to 10 ms later at TRAcT's 100 Hz.  Held parameters make the length of a "period" free
 * (nothing is interpolated over it; the tube's sample rate stays the one the control rate and tube length derive,
 * tube.c:596-612): with a slice of sampleRate / 1000 samples a write lands within a millisecond (shim/tract_tube.c does that).
 * `tube_samples` >= 4, or 0 for the control period.  Only between utterances; TRM_EINVAL otherwise. */
int  trm_stream_set_slice(trm_stream *stream, uint32_t tube_samples);
uint32_t trm_stream_slice(const trm_stream *stream);
int  trm_stream_kernel(const trm_stream *stream);          /* TRM_KERNEL_WIDE or TRM_KERNEL_QUAD (below) */
int  trm_stream_mode(const trm_stream *stream);
/* Exact number of samples per voice the next push of `nframes` frames (resp. the finish call) returns. */
size_t trm_stream_samples_for_push(const trm_stream *stream, size_t nframes);
size_t trm_stream_samples_for_finish(const trm_stream *stream);
/* frames: fp32 [nvoices][nframes][16], host.  The first frame ever pushed is the utterance's starting point;
 * every later frame adds one control period (interpolated from the frame before it, TRMTubeModel.m:611-688).
 * out: fp32 [nvoices][out_pitch] host, out_pitch >= trm_stream_samples_for_push; *nout = samples per voice;
 * max_out[v] (optional) = max |sample| of voice v in this chunk. */
int  trm_stream_push(trm_stream *stream, const float *frames, size_t nframes, float *out, size_t out_pitch,
                     uint32_t *nout, float *max_out);
/* The converter's flush (TRMSampleRateConverter.m:155-168): the last samples of the utterance.  The stream
 * can then start a new utterance with its next push. */
int  trm_stream_finish(trm_stream *stream, float *out, size_t out_pitch, uint32_t *nout, float *max_out);

/* Device-buffer forms: d_frames fp32 [nvoices][nframes][16] and d_out fp32 [nvoices][out_pitch] are HIP device pointers on
 * the stream's device, d_max_out (optional) fp32 [nvoices]; the calls are asynchronous on `stream` (a hipStream_t or NULL)
 * and nothing crosses PCIe -- what a server that mixes, encodes or plays the voices on the device uses, and what the
 * number of concurrent real-time voices is measured with (tools/realtime_voices.py): PCM returned to the host costs
 * 88-176 KB per voice-second, so PCIe (not the kernel) bounds a host-returning stream at ~0.3-0.6 M voices per GPU.
 * *nout is known on return (it depends on the frame count only).  Host- and device-buffer calls of one stream may be mixed,
 * and successive calls may name different HIP streams: a chunk is ordered behind the one before it on the device (an event).
 * The host waits only when a chunk's shape (frames per push, out_pitch) changes or the noise sequence has to grow.
 * (The kernels store PCM in 128-byte pieces: a 128-byte aligned d_out and an out_pitch that is a multiple of 32 floats keep every
 * piece within one cache line.) */
int  trm_stream_push_device(trm_stream *stream, const float *d_frames, size_t nframes, float *d_out, size_t out_pitch,
                            uint32_t *nout, float *d_max_out, void *hip_stream);
int  trm_stream_finish_device(trm_stream *stream, float *d_out, size_t out_pitch, uint32_t *nout, float *d_max_out,
                              void *hip_stream);

/* Kernel form of the synthesis launch.  All forms compute the same samples (same arithmetic per value);
 * they differ in how a voice is laid out on the machine:
 *   TRM_KERNEL_WIDE  one voice per lane, 64 voices per workgroup: highest throughput once the batch fills
 *                    the chip (AUTO: above 32 voices per CU, 8192 on MI355X);
 *   TRM_KERNEL_QUAD  four lanes per voice, 16 voices per workgroup: mid-size batches (AUTO: above 16 voices per CU)
 *                    and every stream (trm_stream_*).  One-shot batches need a control period of at least 24 tube
 *                    samples (the control frames are staged in LDS a period ahead); TRM_KERNEL_WIDE runs otherwise;
 *   TRM_KERNEL_OCT   eight lanes per voice, 8 voices per workgroup, two workgroups per CU: lowest latency for
 *                    batches of up to 16 voices per CU (4096 on MI355X).  Needs a control period of at least 16 tube
 *                    samples; longer batches run TRM_KERNEL_QUAD, shorter periods TRM_KERNEL_WIDE instead.
 * TRM_KERNEL_AUTO (default) picks by batch size; the environment variable TRM_TUBE_KERNEL=wide|quad|oct
 * overrides AUTO (diagnostics).  Parameters with more than four output samples per tube sample (96 kHz output)
 * always run TRM_KERNEL_WIDE.  No reference counterpart: the reference runs one tube per thread. */
enum { TRM_KERNEL_AUTO = 0, TRM_KERNEL_WIDE = 1, TRM_KERNEL_QUAD = 2, TRM_KERNEL_OCT = 3 };
int  trm_batch_set_kernel(trm_batch *batch, int kernel);
int  trm_batch_last_kernel(const trm_batch *batch);

/* Time split.  -[TRMTubeModel synthesize] (TRMTubeModel.m:272-361) is a serial recurrence per voice, so a batch of a few
 * long utterances (GnuTTSServer's one tube per sentence, PhoneToSpeech.m:66-88) lasts as long as its longest voice
 * whatever the machine.  But the tube forgets: every travelling wave is multiplied by dampingFactor = 1 - lossFactor/100
 * once per sample (:216), the end filters, throat and frication band-pass are stable filters, the oscillator FIR and the
 * converter are feed-forward, the noise is a fixed sequence and the oscillator position an exact prefix sum.  A time-split
 * launch cuts every utterance into segments of `periods` control periods (the first one a warm-up longer) and runs them side
 * by side -- a workgroup is one segment of 64 voices in the one-voice-per-lane form, of 16 in the four-lane form (small
 * batches, a single utterance: 1 s of speech in 0.6 ms); trm_batch_last_kernel names the form -- each from rest a
 * warm-up ahead of its first period; the warm-up is chosen by the library so that 1e-5 of the forgotten state is left
 * (damping^W <= 1e-5: 30 control periods at Monet's defaults; measured against the oracle in tools/timesplit_study.py
 * and by the parity tests at the one tolerance, 1e-5).  numberSamples and the sample positions are exact as always.
 *   TRM_TIME_SPLIT_AUTO (default)  the library splits when its launch-time model says so (a form set by name with
 *                                  trm_batch_set_kernel / TRM_TUBE_KERNEL runs whole utterances);
 *   TRM_TIME_SPLIT_OFF             whole utterances always;
 *   periods > 0                    segments of that many control periods (TRM_ERANGE when the tube never forgets:
 *                                  lossFactor 0).
 * Down-sampling batches (tube rate above the output rate) split the same way: the segments write their stretches of the
 * tube-rate rows and the down-sampling kernel converts them as ever.  A control track whose frication bandwidth
 * falls below what the warm-up covers (some tens of Hz; Monet's minimum is 250) is found on the device before the launch
 * and the batch then runs as whole utterances -- the call stays asynchronous either way.  The environment variable
 * TRM_TIME_SPLIT=off|auto|<periods>, read when a batch object is created, sets the default (diagnostics, tests).
 * trm_batch_last_time_split: what the last launch was set up with (periods 0 = whole utterances).
 * No reference counterpart. */
enum { TRM_TIME_SPLIT_AUTO = -1, TRM_TIME_SPLIT_OFF = 0 };
int  trm_batch_set_time_split(trm_batch *batch, int periods);
int  trm_batch_last_time_split(const trm_batch *batch, uint32_t *periods, uint32_t *warm_periods);
/* A ragged batch through the device-buffer entry: the library sees the lengths (d_nframes) only on the device, and AUTO then
 * sizes the segments as if every voice were as long as the longest.  trm_batch_hint_frames hands it a host copy of the
 * nframes array (in the launch's voice order) for the NEXT trm_batch_synthesize_device call: AUTO then counts the workgroups
 * that have work (a block of 64 voices x the segments its longest voice reaches) and picks shorter segments for a batch whose
 * voices mostly end early -- the GnuTTSServer sentence batch 1.8 ms instead of 2.3.  The host-buffer entries do this themselves.
 * Results do not depend on the hint (any split agrees with whole utterances to 1e-5); nframes == NULL withdraws it. */
int  trm_batch_hint_frames(trm_batch *batch, const uint32_t *nframes, size_t nvoices);

/* Average device time (ms) of the tube kernel launches since the last call, measured
 * with hipEvents on the launch stream; resets the accumulator.  Used by bench.py. */
int  trm_batch_kernel_time_ms(trm_batch *batch, double *total_ms, uint32_t *launches);
/* Launch timing on (default) / off.  Off, trm_batch_synthesize_device creates no events and queries none: the call is
 * then only stream work (once the batch has synthesized an utterance at least as long, so that its tables are in place)
 * and can be captured into a HIP graph (hipStreamBeginCapture / torch.cuda.graph) and replayed. */
int  trm_batch_set_timing(trm_batch *batch, int on);

/* Diagnostic: copies the first n entries of the device-resident low-passed noise sequence
 * (TRMUtility.m:71-85 + TRMFilters.m:81-86, generated on the GPU in fp64, stored fp32) to host. */
int  trm_batch_noise_table(trm_batch *batch, float *host_out, size_t n);

/* Library / device identification. */
int  trm_device_count(void);
const char *trm_build_info(void);
/* Diagnostic: resident workgroups (64 voices each) of the tube kernel per CU, per the HIP occupancy query. */
int  trm_kernel_blocks_per_cu(void);
/* Same for a given kernel form (TRM_KERNEL_WIDE / TRM_KERNEL_QUAD). */
int  trm_kernel_blocks_per_cu_form(int kernel);

#ifdef __cplusplus
}
#endif
#endif /* TRM_C_API_H */
