// trm_kernels.h -- launch interface between the C-ABI host code and trm_kernels.hip.
#pragma once

#include <hip/hip_runtime_api.h>
#include <stdint.h>

#include "../../include/trm_c_api.h"
#include "trm_lane.h"

namespace trm {

// Device pointers of one batch launch.  Layout in HBM:
//   frames        fp32 [sum nframes][16], voice v owns rows frame_offset[v] .. +nframes[v]
//   out           fp32 PCM at output rate, voice v's samples at out + out_offset[v]
//   lp_noise      fp32 [>= max tube samples + 2*pad + 256]: the voice-independent low-passed noise sequence
//   src_rows      fp32 [65536][16]: converter coefficients per 16-bit phase (13 used)
//   sine          fp32 [512]
struct TubeArgs {
    const float *frames;
    const uint64_t *frame_offset;
    const uint32_t *nframes;
    float *out;
    const uint64_t *out_offset;
    uint32_t *number_samples;
    float *max_sample;
    const float *lp_noise;
    const float *src_rows;
    const float *sine;
    // down-sampling batches only (tube rate above the output rate): the tube stage writes its tube-rate
    // samples here (voice v at tube_out + tube_offset[v], ntube[v] + 2*pad floats incl. the zero flush) and
    // trm_downsample_kernel converts them; null otherwise
    float *tube_out;
    const uint64_t *tube_offset;
    uint32_t nvoices;
    uint32_t wg_base = 0;         // trm_tube_kernel only: index of the launch's first workgroup within the batch (set by launch_tube)
    uint32_t max_nframes;         // the host sized the noise table (and the tube rows) for this many frames per voice:
                                  // a longer nframes[v] is cut to it (a caller's mistake must not run past them)
    unsigned long long *stamps;   // diagnostic builds only (TRM_STAMP); null in the product
    // Streaming (trm_tube_kernel_q only): a chunk of a longer utterance.  Null for one-shot synthesis.
    //   stream_state   kStreamFloats floats per voice, carried from one chunk to the next
    //   stream_flags   bit 0: first chunk (state ignored: the tube starts at rest, the converter with its 25 zeros
    //                  of pre-roll); bit 1: last chunk (the converter's 2*pad zeros of flush are appended)
    //   stream_n_base  tube samples synthesized before this chunk; stream_k_base / stream_k_end: the chunk emits
    //                  converter outputs k_base <= k < k_end (global indices).  Same for every voice of the launch.
    float *stream_state;
    uint32_t stream_flags, stream_n_base, stream_k_base, stream_k_end;
    // Time-split launches (trm_tube_kernel<kModeSegments> only; seg_periods == 0 otherwise).  An utterance is cut every
    // seg_periods control periods; workgroup w runs segment w / seg_wg_per_seg of the voices 64 * (w % seg_wg_per_seg) ..+63
    // from REST, seg_warm control periods before the segment's first one (the tube forgets like its slowest pole:
    // trm_capi.cc plan_time_split), and emits the converter outputs whose read position lies in the segment proper.  What a
    // segment cannot reconstruct from the frames is the oscillator position: seg_phase[q * 64 * seg_wg_per_seg + v] is the
    // (wrapped) advance of voice v's oscillator between the warm-up starts of segments q - 1 and q (trm_phase_segment_kernel);
    // (seg_phase's row pitch is seg_wg_per_seg x the voices of a workgroup: 64 here, 16 in trm_tube_kernel_q's segment instance)
    // the kernel sums q = 1 .. its own segment (exact sums: osc_increment).  max_sample is folded with an atomic max
    // (zeroed by the launcher), number_samples written by segment 0.
    uint32_t seg_periods = 0, seg_warm = 0, seg_wg_per_seg = 0;
    uint32_t seg_first = 0;           // control periods of segment 0 (= seg_periods + seg_warm: it needs no warm-up, so it is that much
                                      // longer and every workgroup runs the same number of periods); segment s >= 1 starts at
                                      // seg_first + (s - 1) * seg_periods
    uint32_t seg_grid = 0;            // workgroups of the launch: seg_wg_per_seg * (segments of the longest voice)
    const double *seg_phase = nullptr;
    // Device-side choice between two launches of one batch (time-split vs whole utterances): a kernel with a gate returns
    // at once unless (*gate != 0) == gate_want.  Null: no gate.
    const uint32_t *gate = nullptr;
    uint32_t gate_want = 0;
    // Time-split launches: workgroup w of the grid runs segment seg_map[w].x of the block of voices seg_map[w].y, the pairs
    // that have work first (trm_seg_map_kernel).  Null: w / seg_wg_per_seg and w % seg_wg_per_seg.
    const uint2 *seg_map = nullptr;
};

// trm_phase_*_kernel: the oscillator advances a time-split launch starts from, and the guard that decides whether the batch
// may be split at all: *gate is set when a frame's frication bandwidth lies below bw_floor (the band-pass then remembers
// longer than the warm-up; whole-utterance launch instead).  `nseg` = segments of the longest voice.
struct PhaseArgs {
    const float *frames;
    const uint64_t *frame_offset;
    const uint32_t *nframes;
    double *period_adv;           // scratch, nvoices * max_nframes doubles: the oscillator's advance per (voice, control period)
    double *seg_phase;
    uint32_t *gate;
    float bw_floor;
    uint32_t nvoices, max_nframes, nseg, seg_periods, seg_warm, seg_wg_per_seg, seg_first;
    uint32_t voices_per_wg;       // of the tube kernel that follows: 64 (trm_tube_kernel) or 16 (trm_tube_kernel_q)
    // the launch order of the (segment, block of voices) pairs: those with work first (null: none is built)
    uint2 *seg_map = nullptr;
    uint32_t *block_frames = nullptr;     // scratch, seg_wg_per_seg entries: the longest voice of every block
};
hipError_t launch_phase(const Const &c, const PhaseArgs &a, hipStream_t stream);

struct ScaleArgs {
    const float *pcm;
    const uint64_t *out_offset;
    const uint32_t *number_samples;
    const float *max_sample;
    int16_t *pcm16;
    double volumeAmp;     // amplitude(volume)
    double balance;
    int32_t channels;
    int32_t forWavData;
};

int tube_kernel_blocks_per_cu();
hipError_t launch_noise(float *lp, uint32_t from, uint32_t to, double *state, hipStream_t stream);
hipError_t launch_tube(const Const &c, const TubeArgs &a, hipStream_t stream);
// small-batch form (trm_quad.hip): 16 voices per workgroup, four lanes per voice
constexpr int kStreamFloats = 192;   // oscillator position, filter memories, 32 samples of FIR / converter history, 4 x 20 tube values
// `cus` = the device's compute units: more workgroups than that run the instance that fits two per CU
hipError_t launch_tube_quad(const Const &c, const TubeArgs &a, hipStream_t stream, int cus);
// eight lanes per voice, 8 voices per workgroup (trm_oct.hip): one-shot batches of at most two workgroups per CU
hipError_t launch_tube_oct(const Const &c, const TubeArgs &a, hipStream_t stream);
int tube_oct_kernel_blocks_per_cu();
int tube_quad_kernel_blocks_per_cu(int sub);     // sub = blocks per pipeline step of the instance asked about (1 or 2)
// Down-sampling converter (TRMSampleRateConverter.m:234-297) over tube-rate samples in HBM.
struct DownArgs {
    const float *tube;            // tube-rate samples incl. 2*pad zeros of flush per voice
    const uint64_t *tube_offset;
    const uint32_t *nframes;
    float *out;
    const uint64_t *out_offset;
    uint32_t *number_samples;
    float *max_sample;
    const float *fine;            // fine[q] = h[q>>8] + deltaH[q>>8]*(q&255)/256, q < 3328*256
    uint32_t nvoices;
    uint32_t max_nframes;         // as in TubeArgs
    // per-phase coefficient rows (trm_setup.h: build_down_rows); null / too wide for LDS: the generic kernel walks `fine`
    const float *rows;
    uint32_t lmax, rmax, pitch;
    // Streaming (tiled kernel only): this launch converts a chunk of a longer utterance.  tube + tube_offset[v] then
    // points at global tube sample n_origin (the chunk's history first), samples n_origin <= n < n_hi exist, and the
    // launch emits outputs k_base <= k < k_end (global indices) at out + out_offset[v] + (k - k_base).  One-shot:
    // stream = 0 and the bounds come from nframes.
    int stream;
    long long n_origin, n_hi;
    uint32_t k_base, k_end;
    uint32_t ntiles = 0;          // tiled kernel: time tiles of the launch (set by launch_downsample)
};
hipError_t launch_downsample(const Const &c, const DownArgs &a, hipStream_t stream);
// whether the tiled kernel (the only one that converts stream chunks) can run a converter of lmax + rmax taps
bool downsample_tiled_fits(const Const &c, uint32_t lmax, uint32_t rmax);
hipError_t launch_int16(const ScaleArgs &s, uint32_t nvoices, hipStream_t stream);
// sound-file images (header + int16 payload in the container's byte order) per voice, on the device
struct FileArgs {
    ScaleArgs s;                  // (pcm16 unused)
    uint8_t *files;
    const uint64_t *file_offset;  // byte offset of voice v's image
    int32_t format;               // TRM_SOUND_FILE_FORMAT_AU / _AIFF / _WAVE
    uint8_t header[56];           // the container's header with its size fields left 0 (trm_io.cc: io_sound_file_header)
};
hipError_t launch_file_images(const FileArgs &f, uint32_t nvoices, hipStream_t stream);
// out[v * pitch + i] *= g (i < count), mx[v] *= g
hipError_t launch_gain(float *out, size_t pitch, uint32_t count, uint32_t nvoices, float *mx, float g, hipStream_t stream);

// Control-track generation (trm_tracks.hip): event lists -> 16-column frames, one wave per utterance.
struct TrackArgs {
    const uint32_t *event_times;      // u32 [sum nevents]
    const double *event_values;       // f64 [sum nevents][36], NaN = absent
    const uint64_t *event_offset;     // first event of voice v
    const uint32_t *nevents;
    float *frames;                    // f32 [sum nframes][16]
    const uint64_t *frame_offset;     // first frame row of voice v
    uint32_t *nframes_out;
    trm_intonation settings;
    uint32_t nvoices;
};
hipError_t launch_tracks(const TrackArgs &a, hipStream_t stream);

}  // namespace trm
