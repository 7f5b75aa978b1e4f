// trm_capi.cc -- implementation of include/trm_c_api.h on top of the HIP kernels.
//
// The product path is HIP only: if no gfx950 device is usable every synthesis entry point
// fails with TRM_ENODEVICE / TRM_EHIP.  There is no CPU fallback here and nothing in this
// library links or loads oracle/.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <mutex>
#include <string>
#include <utility>
#include <thread>
#include <algorithm>
#include <vector>

#include "../../include/trm_c_api.h"
#include "trm_io.h"
#include "trm_kernels.h"
#include "trm_setup.h"

namespace {

thread_local std::string g_err;
#ifdef TRM_STAMP
unsigned long long *g_stampPtr = nullptr;
#endif

int fail(int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define HIP_TRY(expr)                                                                         \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess) return fail(TRM_EHIP, "%s: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

template <class T>
struct DevBuf {
    T *p = nullptr;
    size_t cap = 0;   // elements
    ~DevBuf() { if (p) (void)hipFree(p); }
    int reserve(size_t n)
    {
        if (n <= cap) return TRM_OK;
        if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
        size_t want = n + n / 4 + 64;
        hipError_t e = hipMalloc((void **)&p, want * sizeof(T));
        if (e != hipSuccess) return fail(TRM_EHIP, "hipMalloc(%zu bytes): %s", want * sizeof(T), hipGetErrorString(e));
        cap = want;
        return TRM_OK;
    }
};

}  // namespace

struct trm_batch {
    trm_input_params params;
    trm::Const c;
    trm_derived d;
    int device = 0;
    hipStream_t stream = nullptr;        // used by the host-buffer entry points
    trm::Const *dConst = nullptr;
    // read-only tables shared per process and device (trm_batch_create): not owned
    const float *dRows = nullptr, *dSine = nullptr;
    const float *dFine = nullptr;        // down-sampling batches only
    const float *dDownRows = nullptr;    // down-sampling batches only: per-phase coefficient rows
    uint32_t downL = 0, downR = 0, downPitch = 0;
    DevBuf<float> dTube;                 // down-sampling: tube-rate samples between the two kernels
    DevBuf<uint64_t> dTubeOff;
    DevBuf<float> dNoise;
    double *dNoiseState = nullptr;
    uint32_t noiseLen = 0;
    // host-form staging
    DevBuf<float> dFrames, dOut, dMax;
    DevBuf<int16_t> dOut16;
    // trm_batch_generate_frames_host staging
    DevBuf<uint32_t> evT, evN;
    DevBuf<double> evV;
    DevBuf<uint64_t> evOff;
    DevBuf<float> evF;
    DevBuf<uint64_t> dFrameOff, dOutOff;
    DevBuf<uint32_t> dNFrames, dNSamples;
    // kernel timing (hipEvents on the launch stream)
    std::vector<std::pair<hipEvent_t, hipEvent_t>> events;      // launches not yet folded into the sums below
    double timedMs = 0.0;
    uint32_t timedLaunches = 0;
    bool timing = true;
    int kernel = TRM_KERNEL_AUTO;        // trm_batch_set_kernel
    uint32_t wideThreshold = 4097;       // voices from which the one-voice-per-lane kernel is the faster form (set at create)
    int lastKernel = TRM_KERNEL_AUTO;    // what the last launch ran
    int cus = 0;                         // compute units of the device (set at create)
    int envKernel = TRM_KERNEL_AUTO;     // TRM_TUBE_KERNEL, read once at create (steers launches left on AUTO; tests)
    bool envDownGeneric = false;         // TRM_DOWNSAMPLE_GENERIC, read once at create (tests: the generic down-sampling kernel)
    size_t tubeOffVoices = 0;            // dTubeOff holds pitch * v for v < tubeOffVoices ...
    uint64_t tubeOffPitch = 0;           // ... at this row pitch (down-sampling batches: rebuilt only when either changes)
    // time-split launches (trm_batch_set_time_split)
    int splitSetting = TRM_TIME_SPLIT_AUTO;      // AUTO, OFF, or a segment length in control periods
    uint32_t lastSplitPeriods = 0, lastSplitWarm = 0;      // what the last launch did (0: whole utterances)
    int lastSplitForm = TRM_KERNEL_WIDE;
    DevBuf<double> dSegPhase, dPeriodAdv;
    DevBuf<uint2> dSegMap;               // a time-split grid's launch order (trm_seg_map_kernel)
    DevBuf<uint32_t> dBlockFrames;
    uint32_t *dGate = nullptr;
    uint64_t hintTotalPeriods = 0;       // set by the host-buffer entries (they see every voice's length) for the launch that follows
    std::vector<uint32_t> hintFrames;    // every voice's frame count in launch order (trm_batch_hint_frames / the host entries), for that launch
};

struct trm_tube {
    trm_batch *b = nullptr;
    std::vector<float> samples;
    uint32_t numberSamples = 0;
    float maxSample = 0.f;
};

extern "C" {

const char *trm_strerror(int code)
{
    switch (code) {
    case TRM_OK: return "ok";
    case TRM_EINVAL: return "invalid argument";
    case TRM_EINVAL_LENGTH: return "illegal tube length";
    case TRM_EFIR: return "oscillator FIR design failed";
    case TRM_ENOMEM: return "out of memory";
    case TRM_EHIP: return "HIP runtime error";
    case TRM_ENODEVICE: return "no usable gfx950 device";
    case TRM_EIO: return "file i/o error";
    case TRM_EPARSE: return "truncated input file";
    case TRM_ESILENT: return "maximum sample value is zero";
    case TRM_ERANGE: return "parameters outside the supported range";
    }
    return "unknown";
}

const char *trm_last_error(void) { return g_err.c_str(); }

int trm_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

const char *trm_build_info(void) { return "libtrm_hip gfx950 (one tube per lane, wave64) " __DATE__; }

int trm_kernel_blocks_per_cu(void) { return trm::tube_kernel_blocks_per_cu(); }
int trm_kernel_blocks_per_cu_form(int kernel)
{
    return kernel == TRM_KERNEL_QUAD ? trm::tube_quad_kernel_blocks_per_cu(1) : trm::tube_kernel_blocks_per_cu();
}

void trm_free(void *p) { free(p); }

// ------------------------------------------------------------------ batch object
int trm_batch_create(const trm_input_params *params, int device, trm_batch **out)
{
    if (!params || !out) return fail(TRM_EINVAL, "null argument");
    *out = nullptr;
    trm::Const c;
    trm_derived d;
    int rc = trm::build_const(*params, c, d);
    if (rc != TRM_OK) {
        if (rc == TRM_EINVAL_LENGTH) fprintf(stderr, "Illegal tube length: %g\n", params->length);   // TRMTubeModel.m:205
        return fail(rc, "%s", trm_strerror(rc));
    }
    if (c.controlPeriod < 4)
        return fail(TRM_ERANGE, "control period of %d tube samples is below the kernel's pipeline step", c.controlPeriod);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(TRM_ENODEVICE, "no HIP device visible");
    if (device < 0) {
        if (hipGetDevice(&device) != hipSuccess) device = 0;
    }
    if (device >= ndev) return fail(TRM_EINVAL, "device %d out of range (%d visible)", device, ndev);
    HIP_TRY(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(TRM_ENODEVICE, "device %d is %s; libtrm_hip carries gfx950 code only", device, prop.gcnArchName);

    trm_batch *b = new (std::nothrow) trm_batch();
    if (!b) return fail(TRM_ENOMEM, "trm_batch");
    b->params = *params;
    b->c = c;
    b->d = d;
    b->device = device;
    // The four-lane form wins up to two rounds of one workgroup (16 voices) per CU: one workgroup keeps a CU's
    // four SIMDs busy, so more voices run in rounds (measured on 256 CUs: 4096 voices 3.1 ms, 8192 6.2 ms, 12288
    // 9.0 ms) while the one-voice-per-lane form takes 7.1 ms for anything up to 16384 (profiles/sweep_forms_r01.txt).
    b->wideThreshold = 2u * 16u * (uint32_t)prop.multiProcessorCount + 1u;
    b->cus = prop.multiProcessorCount;
    // (tests and experiments: 0 = the four-lane form always with two blocks per pipeline step, 1 = always with one --
    // tests/test_gpu_parity.py runs every parity test in that instance too)
    if (const char *e = getenv("TRM_QUAD_CUS")) b->cus = atoi(e);
    if (const char *e = getenv("TRM_TUBE_KERNEL")) b->envKernel = !strcmp(e, "wide") ? TRM_KERNEL_WIDE : !strcmp(e, "quad") ? TRM_KERNEL_QUAD : !strcmp(e, "oct") ? TRM_KERNEL_OCT : TRM_KERNEL_AUTO;
    b->envDownGeneric = getenv("TRM_DOWNSAMPLE_GENERIC") != nullptr;
    hipError_t e;
#define B_TRY(expr)                                                              \
    if ((e = (expr)) != hipSuccess) {                                            \
        trm_batch_destroy(b);                                                    \
        return fail(TRM_EHIP, "%s: %s", #expr, hipGetErrorString(e));            \
    }
    B_TRY(hipStreamCreate(&b->stream));
    B_TRY(hipMalloc((void **)&b->dConst, sizeof(trm::Const)));
    B_TRY(hipMemcpy(b->dConst, &b->c, sizeof(trm::Const), hipMemcpyHostToDevice));
    B_TRY(hipMalloc((void **)&b->dNoiseState, 2 * sizeof(double)));
    B_TRY(hipMalloc((void **)&b->dGate, sizeof(uint32_t)));
    if (const char *e = getenv("TRM_TIME_SPLIT"))          // off | auto | <control periods per segment> (tests, experiments)
        b->splitSetting = !strcmp(e, "off") ? TRM_TIME_SPLIT_OFF : !strcmp(e, "auto") ? TRM_TIME_SPLIT_AUTO : atoi(e);
    {
        // Read-only tables: built and uploaded once per process and device (per ratio for the down-sampling rows) and
        // shared by every batch object there -- a fresh tube per utterance (TRMSynthesizer.m:118-136) finds them in place.
        // They live as long as the process.
        struct DeviceTables {
            float *rows = nullptr, *sine = nullptr, *fine = nullptr;
            struct Down { uint32_t phaseIncrement; double ratio; uint32_t l, r, pitch; float *rows; };
            std::vector<Down> down;
        };
        static std::mutex mu;
        static std::vector<DeviceTables> tables;
        static std::vector<float> hostFine;
        std::lock_guard<std::mutex> lock(mu);
        if (tables.size() < (size_t)ndev) tables.resize(ndev);
        DeviceTables &t = tables[device];
        if (!t.rows) {
            std::vector<float> rows, sine;
            trm::build_src_rows(rows);
            trm::build_sine_table(sine);
            // [4 zeros in front of row 0: the convert stage's shifted fetch of a row reads up to 3 floats before it][rows]
            float *alloc = nullptr;
            B_TRY(hipMalloc((void **)&alloc, (rows.size() + 4) * sizeof(float)));
            B_TRY(hipMemset(alloc, 0, 4 * sizeof(float)));
            B_TRY(hipMemcpy(alloc + 4, rows.data(), rows.size() * sizeof(float), hipMemcpyHostToDevice));
            B_TRY(hipMalloc((void **)&t.sine, sine.size() * sizeof(float)));
            B_TRY(hipMemcpy(t.sine, sine.data(), sine.size() * sizeof(float), hipMemcpyHostToDevice));
            t.rows = alloc + 4;
        }
        b->dRows = t.rows;
        b->dSine = t.sine;
        if (!c.upsample) {
            if (hostFine.empty()) trm::build_src_fine(hostFine);
            if (!t.fine) {
                float *p = nullptr;
                B_TRY(hipMalloc((void **)&p, hostFine.size() * sizeof(float)));
                B_TRY(hipMemcpy(p, hostFine.data(), hostFine.size() * sizeof(float), hipMemcpyHostToDevice));
                t.fine = p;
            }
            b->dFine = t.fine;
            // rows of the tiled down-sampling kernel (launch_downsample falls back to walking `fine` when a row is too wide)
            if (c.phaseIncrement > 0 && trm::kSrcFineLen / c.phaseIncrement <= 160) {
                const DeviceTables::Down *hit = nullptr;
                for (const auto &dn : t.down)
                    if (dn.phaseIncrement == c.phaseIncrement && dn.ratio == c.sampleRateRatioD) hit = &dn;
                if (!hit) {
                    DeviceTables::Down dn;
                    std::vector<float> drows;
                    dn.phaseIncrement = c.phaseIncrement;
                    dn.ratio = c.sampleRateRatioD;
                    trm::build_down_rows(hostFine, dn.ratio, dn.phaseIncrement, dn.l, dn.r, dn.pitch, drows);
                    dn.rows = nullptr;
                    B_TRY(hipMalloc((void **)&dn.rows, drows.size() * sizeof(float)));
                    B_TRY(hipMemcpy(dn.rows, drows.data(), drows.size() * sizeof(float), hipMemcpyHostToDevice));
                    t.down.push_back(dn);
                    hit = &t.down.back();
                }
                b->downL = hit->l; b->downR = hit->r; b->downPitch = hit->pitch;
                b->dDownRows = hit->rows;
            }
        }
    }
#undef B_TRY
    *out = b;
    return TRM_OK;
}

void trm_batch_destroy(trm_batch *b)
{
    if (!b) return;
    (void)hipSetDevice(b->device);
    for (auto &ev : b->events) { (void)hipEventDestroy(ev.first); (void)hipEventDestroy(ev.second); }
    if (b->dConst) (void)hipFree(b->dConst);
    if (b->dNoiseState) (void)hipFree(b->dNoiseState);
    if (b->dGate) (void)hipFree(b->dGate);
    if (b->stream) (void)hipStreamDestroy(b->stream);
    delete b;
}

int trm_batch_derived(const trm_batch *b, trm_derived *out)
{
    if (!b || !out) return fail(TRM_EINVAL, "null argument");
    *out = b->d;
    return TRM_OK;
}

size_t trm_batch_samples_for_frames(const trm_batch *b, size_t nframes)
{
    if (!b || nframes == 0) return 0;                                   // TRMTubeModel.m:274-277
    return (size_t)trm::count_outputs(b->d, (uint64_t)(nframes - 1) * (uint64_t)b->d.controlPeriod);
}

int trm_derive(const trm_input_params *params, trm_derived *out)
{
    if (!params || !out) return fail(TRM_EINVAL, "null argument");
    trm::Const c;
    int rc = trm::build_const(*params, c, *out);
    if (rc) return fail(rc, "%s", trm_strerror(rc));
    return TRM_OK;
}

size_t trm_samples_for_frames(const trm_input_params *params, size_t nframes)
{
    trm_derived d;
    if (nframes == 0 || trm_derive(params, &d) != TRM_OK) return 0;
    return (size_t)trm::count_outputs(d, (uint64_t)(nframes - 1) * (uint64_t)d.controlPeriod);
}

// The voice-independent noise sequence is generated on the device (fp64, one lane) and cached;
// it only ever grows.  `need` = tube samples incl. the flush tail.
static bool quad_ratio_too_high(const trm::Const &c) { return c.upsample && c.timeRegisterIncrement < 65536u / 4u; }

// Launch timing: finished launches (all of them when `wait`) leave the event list for the running sums.
static int fold_events(trm_batch *b, bool wait)
{
    size_t done = 0;
    hipError_t bad = hipSuccess;
    for (auto &ev : b->events) {
        if (wait) {
            if ((bad = hipEventSynchronize(ev.second)) != hipSuccess) break;       // (this pair stays in the list, untouched)
        } else if (hipEventQuery(ev.second) != hipSuccess) break;       // (in stream order: the later ones are not finished either)
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, ev.first, ev.second) == hipSuccess) {
            b->timedMs += ms;
            b->timedLaunches++;
        }
        (void)hipEventDestroy(ev.first);
        (void)hipEventDestroy(ev.second);
        done++;
    }
    b->events.erase(b->events.begin(), b->events.begin() + done);     // the pairs destroyed above leave the list, error or not
    if (bad != hipSuccess) return fail(TRM_EHIP, "hipEventSynchronize: %s", hipGetErrorString(bad));
    return TRM_OK;
}

// The low-passed noise sequence (TRMUtility.m:71-85, TRMFilters.m:81-86) depends on nothing: not on the voice, not on
// the parameters, not on the device.  It is a serial fp64 recurrence (~55 ns per sample on one lane), so the process
// generates each stretch of it once -- on whichever device first needs it -- keeps it on the host, and every batch
// object uploads what it needs: a fresh tube per utterance (TRMSynthesizer.m:118-136) does not pay for it again.
namespace {
struct NoiseCache {
    std::mutex mu;
    std::vector<float> lp;
    double state[2] = {0.7892347, 0.0};                                 // TRMUtility.m:72-77, TRMTubeModel.m:235
};
NoiseCache g_noise;
}  // namespace

static int ensure_noise(trm_batch *b, uint32_t need, hipStream_t stream)
{
    if (need <= b->noiseLen) return TRM_OK;
    const uint32_t newLen = need + need / 2 + 4096;
    if (newLen > b->dNoise.cap) {
        // grow: a fresh buffer, refilled from the host copy below
        HIP_TRY(hipStreamSynchronize(stream));
        int rc = b->dNoise.reserve(newLen);
        if (rc) return rc;
        b->noiseLen = 0;
    }
    const uint32_t to = (uint32_t)b->dNoise.cap;
    std::lock_guard<std::mutex> lock(g_noise.mu);
    uint32_t have = (uint32_t)g_noise.lp.size();
    if (have < to) {
        // extend the process-wide sequence on this device, straight into this object's buffer, and take it home
        HIP_TRY(hipMemcpyAsync(b->dNoiseState, g_noise.state, sizeof g_noise.state, hipMemcpyHostToDevice, stream));
        HIP_TRY(trm::launch_noise(b->dNoise.p, have, to, b->dNoiseState, stream));
        std::vector<float> fresh(to - have);
        double st[2];
        HIP_TRY(hipMemcpyAsync(fresh.data(), b->dNoise.p + have, fresh.size() * sizeof(float), hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipMemcpyAsync(st, b->dNoiseState, sizeof st, hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipStreamSynchronize(stream));
        g_noise.lp.insert(g_noise.lp.end(), fresh.begin(), fresh.end());     // (only what was generated and fetched is remembered)
        g_noise.state[0] = st[0];
        g_noise.state[1] = st[1];
    } else {
        have = to;
    }
    if (b->noiseLen < have) {
        HIP_TRY(hipMemcpyAsync(b->dNoise.p + b->noiseLen, &g_noise.lp[b->noiseLen], (size_t)(have - b->noiseLen) * sizeof(float),
                               hipMemcpyHostToDevice, stream));
        HIP_TRY(hipStreamSynchronize(stream));                          // the host vector may grow (move) once the lock is gone
    }
    b->noiseLen = to;
    return TRM_OK;
}

// ------------------------------------------------------------------ time-split launches
// The waveguide forgets: every travelling wave is multiplied by the damping factor once per sample (TRMTubeModel.m:216,
// :796-829), the end filters, the throat and the frication band-pass are stable one- and two-pole sections, the FIR and
// the converter are feed-forward, the noise sequence is addressed by its index and the oscillator position is an exact
// prefix sum.  A tube started from rest therefore agrees with the uninterrupted one after a warm-up: the difference
// decays like the slowest pole, pole^n.  (Measured against the oracle, tools/timesplit_study.py: with damping^W = 1e-5
// even a voice with mouth and velum closed -- nothing but the damping factor takes energy out -- is back at the
// unsplit path's own error, 2e-6 worst sample; open voices get there in two thirds of the time.)  An utterance can so be cut in
// time and its pieces run side by side: what bounds a small or ragged batch is the serial chain of its longest voice.
namespace {
struct SplitPlan {
    uint32_t periods = 0;         // control periods per segment; 0 = whole utterances
    uint32_t warm = 0;            // warm-up control periods
    bool allBusy = false;         // the caller's lengths say every workgroup of the grid has work: no launch order to build
    int form = TRM_KERNEL_WIDE;   // the segment instance that runs it: one voice per lane (64 voices per workgroup) or four lanes per voice (16)
    float bwFloor = 0.0f;         // frication bandwidths below this need a longer warm-up: the launch falls back (device-side)
};
// ln(1e-5): what is left of the state the warm-up starts without.  Measured (tools/timesplit_study.py, 180 random and
// adversarial voices on the host model): with this warm-up (30 control periods at Monet's defaults) the split path differs from
// the whole-utterance path by its fp32 noise floor -- worst single sample 8.5e-6 of the maximum, RMS 1.5e-6 -- exactly as with
// 1.1x and 1.2x the warm-up; with 0.9x it starts to show (1.1e-5 / 1.6e-6).  The output error is ~0.15 of the bound for the
// voice that forgets slowest (mouth and velum closed).
constexpr double kSplitLogEps = -11.512925464970229;
}  // namespace

// warm-up (tube samples) after which a tube started from rest has forgotten that it was; 0 = never (a pole on the unit circle)
static uint32_t split_warm_samples(const trm::Const &c)
{
    double pole = fabs((double)c.damping);
    const double others[] = {fabs((double)c.mCoeff), fabs((double)c.nCoeff), fabs((double)c.tb1)};
    for (double o : others) pole = o > pole ? o : pole;
    if (!(pole < 0.99999)) return 0;
    const double w = kSplitLogEps / log(pole);
    if (!(w < 1.0e6)) return 0;
    // + the oscillator FIR's 24 samples of history, the converter's 26-sample window and the pipeline's block granularity
    return (uint32_t)ceil(w) + 64u;
}

// Predicted launch time in ms per second of speech at Monet's rates (19 750 tube samples), from the measured figures of the
// kernel forms on 256 CUs (profiles/sweep_forms_r04.txt, split_probe_r04.txt), by workgroups per CU.
// The one-voice-per-lane kernel: a CU holds two workgroups; one per CU runs at 6.7, two at 7.5, and more go in ROUNDS of two
// per CU at 7.75 each -- a round is not cheaper for being partly filled (its workgroups last as long), except that a last
// round of at most one workgroup per CU saves ~0.8 (measured at 2.5 .. 10 workgroups per CU).
static double wide_cost(const trm_batch *b, uint64_t workgroups)
{
    const double cus = b->cus > 0 ? b->cus : 256, x = (double)workgroups / cus;
    if (x <= 1.0) return 6.7;
    if (x <= 2.0) return 7.5;
    const double rounds = ceil(x / 2.0), last = x - 2.0 * (rounds - 1.0);
    return 7.75 * rounds - (last <= 1.0 ? 0.8 : 0.0);
}
static double unsplit_cost(const trm_batch *b, size_t nvoices, int which)
{
    const double cus = b->cus > 0 ? b->cus : 256, vpc = (double)nvoices / cus;
    if (which == TRM_KERNEL_OCT) return vpc <= 4 ? 2.06 : vpc <= 8 ? 2.24 : 2.37 * (vpc <= 16 ? 1.0 : vpc / 16.0);
    if (which == TRM_KERNEL_QUAD) return vpc <= 16 ? 3.1 : vpc <= 32 ? 4.1 : 4.1 * vpc / 32.0;
    return wide_cost(b, (nvoices + 63) / 64);
}

// segments of an utterance of P control periods cut every `periods`: the first one is periods + warm long (it has no warm-up
// of its own: with it that long every workgroup of the launch runs periods + warm control periods)
static uint32_t split_segments(uint32_t P, uint32_t periods, uint32_t warm)
{
    const uint32_t first = periods + warm;
    return P <= first ? 1u : 1u + (P - first + periods - 1) / periods;
}

// workgroups of a time-split launch that have work: block by block (its longest voice, in control periods) the segments it
// reaches -- what trm_seg_map_kernel counts on the device
static uint64_t busy_workgroups(const std::vector<uint32_t> &longest, uint32_t periods, uint32_t warm)
{
    uint64_t n = 0;
    for (uint32_t per : longest) n += split_segments(per, periods, warm);
    return n;
}

// `which` = the kernel form the launch would take unsplit.  `totalPeriods` = the control periods of all voices together where
// the caller knows them (the host-buffer entries; 0: every voice is taken to be as long as the longest).
static int plan_time_split(const trm_batch *b, size_t nvoices, uint32_t max_nframes, int which, int byName, uint64_t totalPeriods, SplitPlan &pl)
{
    pl = SplitPlan();
    const int setting = b->splitSetting;
    if (setting == TRM_TIME_SPLIT_OFF || max_nframes < 2) return TRM_OK;
    const uint32_t CP = (uint32_t)b->c.controlPeriod, P = max_nframes - 1;
    // the longest voice (in control periods) of every block of 64 / 16 voices, where the caller told the lengths
    std::vector<uint32_t> longest64, longest16;
    if (b->hintFrames.size() == nvoices && setting <= 0) {
        longest64.assign((nvoices + 63) / 64, 0);
        longest16.assign((nvoices + 15) / 16, 0);
        for (size_t v = 0; v < nvoices; v++) {
            const uint32_t nfr = b->hintFrames[v] < max_nframes ? b->hintFrames[v] : max_nframes, per = nfr > 0 ? nfr - 1 : 0;
            longest64[v / 64] = longest64[v / 64] > per ? longest64[v / 64] : per;
            longest16[v / 16] = longest16[v / 16] > per ? longest16[v / 16] : per;
        }
    }
    const uint32_t ws = split_warm_samples(b->c);
    if (ws == 0) {
        if (setting > 0) return fail(TRM_ERANGE, "time split: the tube never forgets (loss factor %g %%)", b->params.lossFactor);
        return TRM_OK;
    }
    const uint32_t warm = (ws + CP - 1) / CP;
    uint32_t periods = 0;
    // the four-lane segment instance: up-sampling batches whose converter makes at most four outputs per tube sample
    const bool quadOk = b->c.upsample && !quad_ratio_too_high(b->c) && which != TRM_KERNEL_WIDE;
    if (setting > 0) {
        periods = (uint32_t)setting;
        // a split asked for by name: in the form asked for by name (four lanes, or one voice per lane), else by size
        const uint64_t wgsQ = (uint64_t)split_segments(P, periods, warm) * ((nvoices + 15) / 16);
        pl.form = (quadOk && (byName == TRM_KERNEL_QUAD || (byName == TRM_KERNEL_AUTO && wgsQ <= (uint64_t)(b->cus > 0 ? b->cus : 256)))) ? TRM_KERNEL_QUAD
                                                                                                                                       : TRM_KERNEL_WIDE;
    } else {
        // AUTO.  A time-split launch pays when its workgroups -- one per segment and block of 64 voices, every one of them
        // periods + warm control periods long -- fill the chip's rounds better than whole utterances do (wide_cost): every
        // segment count is priced, the shortest predicted launch taken when it beats whole utterances by a tenth.
        const double whole = unsplit_cost(b, nvoices, which) * P * CP / 19750.0;
        double best = whole * 0.9;
        // (segments of at least half a warm-up: a launch never does more than three times the arithmetic of whole
        // utterances -- shorter ones still shorten a nearly empty chip's launch a little, at ten times the work)
        uint32_t minPeriods = (255u + CP) / CP > 4u ? (255u + CP) / CP : 4u;
        minPeriods = minPeriods > (warm + 1) / 2 ? minPeriods : (warm + 1) / 2;
        for (uint32_t nseg = 2; nseg <= 4096 && P > warm; nseg++) {
            const uint32_t sp = (P - warm + nseg - 1) / nseg;          // the shortest segments that make `nseg` of them
            if (sp < minPeriods) break;
            const uint64_t segs = split_segments(P, sp, warm);
            if (segs < 2) continue;
            // workgroups with work: per segment the blocks of 64 voices that reach it (they are launched first:
            // trm_seg_map_kernel) -- counted where the caller's lengths are known, else every block in every segment
            const uint64_t wgs = !longest64.empty() ? busy_workgroups(longest64, sp, warm)
                                 : totalPeriods == 0 ? segs * ((nvoices + 63) / 64)
                                                     : (totalPeriods + 64ull * sp - 1) / (64ull * sp) + (segs + 1) / 2;    // (+ partly filled last blocks)
            const double t = 0.03 + wide_cost(b, wgs) * (double)(sp + warm) * CP / 19750.0;
            if (t < best) { best = t; periods = sp; pl.form = TRM_KERNEL_WIDE; }
            // ... or in the four-lane form (16 voices x one segment per workgroup, one workgroup per CU at 3.1 ms per second of
            // speech): a handful of voices, a single utterance
            const uint64_t wgsQ = !longest16.empty() ? busy_workgroups(longest16, sp, warm)
                                  : totalPeriods == 0 ? segs * ((nvoices + 15) / 16) : (totalPeriods + 16ull * sp - 1) / (16ull * sp) + (segs + 1) / 2;
            if (quadOk && wgsQ <= (uint64_t)(b->cus > 0 ? b->cus : 256)) {
                const double tq = 0.03 + 3.1 * (double)(sp + warm) * CP / 19750.0;
                if (tq < best) { best = tq; periods = sp; pl.form = TRM_KERNEL_QUAD; }
            }
        }
    }
    if (periods == 0 || split_segments(P, periods, warm) < 2) return TRM_OK;      // one segment is the whole utterance
    {
        const std::vector<uint32_t> &longest = pl.form == TRM_KERNEL_QUAD ? longest16 : longest64;
        pl.allBusy = !longest.empty() && busy_workgroups(longest, periods, warm) == (uint64_t)split_segments(P, periods, warm) * longest.size();
    }
    pl.periods = periods;
    pl.warm = warm;
    {
        // the frication band-pass (TRMFilters.m:9-29) has poles of radius sqrt(2 beta), 2 beta = (1 - t) / (1 + t),
        // t = tan(pi BW / SR): the bandwidth at which the warm-up leaves 1e-5 of its memory
        const double r2 = exp(2.0 * kSplitLogEps / (double)(warm * CP - 64u));
        const double t = (1.0 - r2) / (1.0 + r2);
        pl.bwFloor = (float)((double)b->d.sampleRate * atan(t) / 3.14159265358979323846);
    }
    return TRM_OK;
}

int trm_batch_hint_frames(trm_batch *b, const uint32_t *nframes, size_t nvoices)
{
    if (!b) return fail(TRM_EINVAL, "null batch");
    if (!nframes || nvoices == 0) { b->hintFrames.clear(); return TRM_OK; }
    b->hintFrames.assign(nframes, nframes + nvoices);
    return TRM_OK;
}

int trm_batch_set_time_split(trm_batch *b, int periods)
{
    if (!b) return fail(TRM_EINVAL, "null batch");
    if (periods < TRM_TIME_SPLIT_AUTO) return fail(TRM_EINVAL, "time split: %d", periods);
    b->splitSetting = periods;
    return TRM_OK;
}

int trm_batch_last_time_split(const trm_batch *b, uint32_t *periods, uint32_t *warm_periods)
{
    if (!b) return fail(TRM_EINVAL, "null batch");
    if (periods) *periods = b->lastSplitPeriods;
    if (warm_periods) *warm_periods = b->lastSplitWarm;
    return TRM_OK;
}

int trm_batch_synthesize_device(trm_batch *b, size_t nvoices, const float *d_frames, const uint64_t *d_frame_offset,
                                const uint32_t *d_nframes, uint32_t max_nframes, float *d_out,
                                const uint64_t *d_out_offset, uint32_t *d_number_samples, float *d_max_sample,
                                void *stream_)
{
    if (!b) return fail(TRM_EINVAL, "null batch");
    if (nvoices == 0) return TRM_OK;
    if (!d_frames || !d_frame_offset || !d_nframes || !d_out || !d_out_offset || !d_number_samples || !d_max_sample)
        return fail(TRM_EINVAL, "null device pointer");
    if (nvoices > 0xFFFFFFFFull - 64) return fail(TRM_EINVAL, "too many voices");
    hipStream_t stream = (hipStream_t)stream_;
    HIP_TRY(hipSetDevice(b->device));
    uint64_t ntubeMax = max_nframes > 0 ? (uint64_t)(max_nframes - 1) * (uint64_t)b->d.controlPeriod : 0;
    if (ntubeMax + 64 > 0x7FFFFFFFull) return fail(TRM_ERANGE, "utterance too long");
    // + 2 ring halves of look-ahead that the kernel's LDS-DMA prefetch may touch
    int rc = ensure_noise(b, (uint32_t)ntubeMax + 2u * (uint32_t)b->d.padSize + 256u, stream);
    if (rc) return rc;
    trm::TubeArgs a;
    a.frames = d_frames;
    a.frame_offset = d_frame_offset;
    a.nframes = d_nframes;
    a.out = d_out;
    a.out_offset = d_out_offset;
    a.number_samples = d_number_samples;
    a.max_sample = d_max_sample;
    a.lp_noise = b->dNoise.p;
    a.src_rows = b->dRows;
    a.sine = b->dSine;
    a.nvoices = (uint32_t)nvoices;
    a.max_nframes = max_nframes;
    a.stamps = nullptr;
    a.stream_state = nullptr;
    a.stream_flags = a.stream_n_base = a.stream_k_base = a.stream_k_end = 0;
    a.tube_out = nullptr;
    a.tube_offset = nullptr;
    if (!b->c.upsample) {
        // tube rate above the output rate: tube-rate samples go through HBM to the down-sampling kernel;
        // voice v gets a fixed-pitch row of (max_nframes-1)*controlPeriod + 2*pad floats
        const uint64_t pitch = (ntubeMax + 2ull * (uint64_t)b->d.padSize + 3ull) & ~3ull;      // rows 16-byte aligned
        if ((rc = b->dTube.reserve(pitch * nvoices + 1))) return rc;
        if (b->tubeOffPitch != pitch || b->tubeOffVoices < nvoices) {
            // the row offsets pitch * v: uploaded when the batch shape changes, not per launch -- the entry stays pure
            // stream work (asynchronous, capturable) for every call that repeats a shape
            if ((rc = b->dTubeOff.reserve(nvoices))) return rc;
            std::vector<uint64_t> offs(nvoices);
            for (size_t i = 0; i < nvoices; i++) offs[i] = pitch * i;
            HIP_TRY(hipMemcpy(b->dTubeOff.p, offs.data(), nvoices * sizeof(uint64_t), hipMemcpyHostToDevice));
            b->tubeOffPitch = pitch;
            b->tubeOffVoices = nvoices;
        }
        a.tube_out = b->dTube.p;
        a.tube_offset = b->dTubeOff.p;
    }
#ifdef TRM_STAMP
    {   // diagnostic library only: per-workgroup, per-role {work, wait} cycle sums; read back with
        // trm_batch_noise_table-like copy in tools/stage_profile.py via TRM_STAMP_PTR
        static unsigned long long *dStamps = nullptr;
        if (!dStamps) HIP_TRY(hipMalloc((void **)&dStamps, 65536 * 16 * sizeof(unsigned long long)));
        HIP_TRY(hipMemsetAsync(dStamps, 0, 65536 * 16 * sizeof(unsigned long long), stream));
        a.stamps = dStamps;
        g_stampPtr = dStamps;
    }
#endif
    hipEvent_t e0 = nullptr, e1 = nullptr;
    struct EventGuard {     // the pair belongs to this call until it is handed to the batch's list
        hipEvent_t &a, &b;
        bool armed = true;
        ~EventGuard() { if (armed) { if (a) (void)hipEventDestroy(a); if (b) (void)hipEventDestroy(b); } }
    } guard{e0, e1};
    if (b->timing) {
        HIP_TRY(hipEventCreate(&e0));
        HIP_TRY(hipEventCreate(&e1));
        HIP_TRY(hipEventRecord(e0, stream));
    }
    // Kernel form: one voice per lane (64 voices per workgroup) once that alone puts two workgroups on every CU; below
    // that, several lanes per voice, which advances several tube samples per pass of the instruction streams and spreads
    // a small batch over more CUs: eight lanes (8 voices per workgroup) while two such workgroups per CU hold the
    // batch, four lanes (16 voices per workgroup) from there on.
    // (trm_oct.hip's feed-forward waves step 8 samples at a time and set a control period up while the lanes of the one
    // before are still crossing into it: a control period must hold at least two steps)
    const bool octFits = b->c.controlPeriod >= 16 && b->cus > 0 && (nvoices + 7) / 8 <= 2 * (size_t)b->cus;
    int which = b->kernel;
    if (which == TRM_KERNEL_AUTO) which = b->envKernel;
    if (which == TRM_KERNEL_AUTO) which = nvoices >= (size_t)b->wideThreshold ? TRM_KERNEL_WIDE : octFits ? TRM_KERNEL_OCT : TRM_KERNEL_QUAD;
    if (which == TRM_KERNEL_OCT && !octFits) which = TRM_KERNEL_QUAD;
    // The converter of the forms with several lanes per voice is fed one block of coefficient rows per step by design
    // (two at a push): four outputs per tube sample.  It measured clean to 5.3 and wrong from 5.6 on (the ring laps the
    // converter; tools/fuzz_parity.py at 96 kHz), so above 4 -- 96 kHz output from any adult tube, 64 kHz from 22 cm on --
    // the one-voice-per-lane form runs, whatever was asked for.
    if (which != TRM_KERNEL_WIDE && quad_ratio_too_high(b->c)) which = TRM_KERNEL_WIDE;
    // ... and so it does for control periods below 24 tube samples (four lanes per voice; 16 with eight: control rates
    // above ~0.8 / 1.2 kHz for an adult tube): those one-shot forms stage the control frames in LDS a period ahead, and a
    // period must hold three (two) of their steps
    if (which == TRM_KERNEL_QUAD && b->c.controlPeriod < 24) which = TRM_KERNEL_WIDE;
    // Time split (above): the utterances cut into segments that run side by side in the one-voice-per-lane form.  A form
    // the caller asked for by name is run as asked (whole utterances) unless the split was asked for by name too.
    SplitPlan pl;
    const bool formByName = b->kernel != TRM_KERNEL_AUTO || b->envKernel != TRM_KERNEL_AUTO;
    const int byName = b->kernel != TRM_KERNEL_AUTO ? b->kernel : b->envKernel;
    if (!(formByName && b->splitSetting <= 0) && (rc = plan_time_split(b, nvoices, max_nframes, which, byName, b->hintTotalPeriods, pl))) return rc;
    b->hintTotalPeriods = 0;                          // (a hint holds for one launch)
    b->hintFrames.clear();
    b->lastSplitPeriods = pl.periods;
    b->lastSplitWarm = pl.periods ? pl.warm : 0;
    if (pl.periods) {
        const uint32_t nseg = split_segments(max_nframes - 1, pl.periods, pl.warm);
        const uint32_t perWg = pl.form == TRM_KERNEL_QUAD ? 16u : 64u;
        const uint32_t wgPerSeg = (uint32_t)((nvoices + perWg - 1) / perWg);
        if ((uint64_t)nseg * wgPerSeg > 0x7FFFFFFFull / 64) return fail(TRM_ERANGE, "time split: too many segments");
        if ((rc = b->dSegPhase.reserve((size_t)nseg * wgPerSeg * perWg)) || (rc = b->dPeriodAdv.reserve(nvoices * (size_t)max_nframes)) ||
            (rc = b->dSegMap.reserve((size_t)nseg * wgPerSeg)) || (rc = b->dBlockFrames.reserve(wgPerSeg))) return rc;
        HIP_TRY(hipMemsetAsync(b->dGate, 0, sizeof(uint32_t), stream));
        HIP_TRY(hipMemsetAsync(d_max_sample, 0, nvoices * sizeof(float), stream));
        trm::PhaseArgs ph;
        ph.frames = d_frames; ph.frame_offset = d_frame_offset; ph.nframes = d_nframes;
        ph.period_adv = b->dPeriodAdv.p; ph.seg_phase = b->dSegPhase.p; ph.gate = b->dGate; ph.bw_floor = pl.bwFloor;
        ph.nvoices = (uint32_t)nvoices; ph.max_nframes = max_nframes; ph.nseg = nseg;
        ph.seg_periods = pl.periods; ph.seg_warm = pl.warm; ph.seg_wg_per_seg = wgPerSeg; ph.seg_first = pl.periods + pl.warm;
        ph.voices_per_wg = perWg;
        ph.seg_map = pl.allBusy ? nullptr : b->dSegMap.p; ph.block_frames = b->dBlockFrames.p;
        HIP_TRY(trm::launch_phase(b->c, ph, stream));
        trm::TubeArgs sa = a;
        sa.seg_periods = pl.periods; sa.seg_warm = pl.warm; sa.seg_wg_per_seg = wgPerSeg; sa.seg_grid = nseg * wgPerSeg;
        sa.seg_first = pl.periods + pl.warm;
        sa.seg_phase = b->dSegPhase.p;
        sa.seg_map = ph.seg_map;
        sa.gate = b->dGate; sa.gate_want = 0;
        if (pl.form == TRM_KERNEL_QUAD) HIP_TRY(trm::launch_tube_quad(b->c, sa, stream, b->cus));
        else HIP_TRY(trm::launch_tube(b->c, sa, stream));
        b->lastSplitForm = pl.form;
        // ... and, should the pre-pass have found a track the warm-up does not cover, whole utterances (the launch below
        // returns at once otherwise)
        a.gate = b->dGate; a.gate_want = 1;
        which = TRM_KERNEL_WIDE;
    }
    b->lastKernel = pl.periods ? b->lastSplitForm : which;
    if (which == TRM_KERNEL_OCT)
        HIP_TRY(trm::launch_tube_oct(b->c, a, stream));
    else if (which == TRM_KERNEL_QUAD)
        HIP_TRY(trm::launch_tube_quad(b->c, a, stream, b->cus));
    else
        HIP_TRY(trm::launch_tube(b->c, a, stream));
    if (!b->c.upsample) {
        trm::DownArgs d;
        d.tube = b->dTube.p;
        d.tube_offset = b->dTubeOff.p;
        d.nframes = d_nframes;
        d.out = d_out;
        d.out_offset = d_out_offset;
        d.number_samples = d_number_samples;
        d.max_sample = d_max_sample;
        d.fine = b->dFine;
        d.nvoices = (uint32_t)nvoices;
        d.max_nframes = max_nframes;
        d.rows = b->envDownGeneric ? nullptr : b->dDownRows;     // (tests: the generic kernel must agree bit for bit)
        d.lmax = b->downL; d.rmax = b->downR; d.pitch = b->downPitch;
        d.stream = 0; d.n_origin = d.n_hi = 0; d.k_base = d.k_end = 0;
        HIP_TRY(trm::launch_downsample(b->c, d, stream));
    }
    if (b->timing) {
        hipError_t e = hipEventRecord(e1, stream);
        b->events.emplace_back(e0, e1);                 // (owned by the list from here on, whatever happened)
        guard.armed = false;
        if (e != hipSuccess) return fail(TRM_EHIP, "hipEventRecord: %s", hipGetErrorString(e));
        // a caller that never asks for the time must not pile up events: fold the finished launches into the sums
        if (b->events.size() >= 64) fold_events(b, false);
    }
    return TRM_OK;
}

// ------------------------------------------------------------------ streaming synthesis
struct trm_stream {
    trm_batch *b = nullptr;
    size_t nvoices = 0;
    DevBuf<float> dState, dFrames, dOut, dMax, dLast, dPushed;
    DevBuf<float> dTube, dHist;       // down-sampling streams: [history | chunk] tube-rate rows; the history between chunks
    DevBuf<uint64_t> dTubeOff, dTubeOff0;
    uint32_t hist = 0;                // tube samples of history a chunk's first output may reach back (multiple of 4)
    DevBuf<uint64_t> dFrameOff, dOutOff;
    DevBuf<uint32_t> dNFrames, dNSamples;
    // dLast: [nvoices][16], the frame the next control period starts from; dPushed / dOut: the host-buffer entries' staging
    std::vector<float> hostOut;
    std::vector<uint64_t> hFrameOff, hOutOff, hTubeOff0, hTubeOff;      // the index arrays of the current chunk shape
    std::vector<uint32_t> hNFrames;
    size_t shapeRows = 0, shapePitch = 0, shapeRowPitch = 0;
    bool haveLast = false;            // an utterance is open
    bool first = true;                // no chunk of it has been synthesized yet
    int mode = TRM_STREAM_MODE_FRAMEWORK;
    bool wide = false;                // trm_kernels.hip's streaming instance (one voice per lane) instead of trm_quad.hip's
    uint64_t nBase = 0, kBase = 0;    // tube samples synthesized / converter outputs emitted so far
    int32_t controlPeriod0 = 0;       // the control period the parameters derive (trm_stream_set_slice(.., 0) returns to it)
    // Chunks of one stream are ordered on the device whichever HIP stream each call names (host entries: the object's own,
    // device entries: the caller's): every chunk ends with this event and a chunk on another stream waits for it first.
    hipEvent_t chunkDone = nullptr;
    hipStream_t lastStream = nullptr;
    bool haveChunk = false;
};

int trm_stream_create(const trm_input_params *params, int device, size_t nvoices, trm_stream **out)
{
    if (!params || !out || nvoices == 0) return fail(TRM_EINVAL, "null argument / no voices");
    *out = nullptr;
    trm_batch *b = nullptr;
    int rc = trm_batch_create(params, device, &b);
    if (rc) return rc;
    if (!b->c.upsample && (!b->dDownRows || b->downR > (uint32_t)b->d.padSize || b->downL > (uint32_t)b->d.padSize + 1u ||
                           !trm::downsample_tiled_fits(b->c, b->downL, b->downR))) {
        // (a chunk emits the outputs whose read position lies inside it; their right wing must end there too)
        trm_batch_destroy(b);
        return fail(TRM_ERANGE, "streaming: output rate too far below the tube rate (%d Hz) for the tiled down-sampling kernel", b->d.sampleRate);
    }
    trm_stream *s = new (std::nothrow) trm_stream();
    if (!s) { trm_batch_destroy(b); return fail(TRM_ENOMEM, "trm_stream"); }
    s->b = b;
    s->nvoices = nvoices;
    s->controlPeriod0 = b->c.controlPeriod;
    // The kernel form is the stream's for life (the saved state is laid out for it): one voice per lane once the voices
    // fill the chip (and for what the four-lane form does not convert: more than four outputs per tube sample), four
    // lanes per voice below that.  TRM_TUBE_KERNEL=wide|quad overrides (diagnostics).
    s->wide = nvoices >= (size_t)b->wideThreshold || quad_ratio_too_high(b->c);
    if (b->envKernel == TRM_KERNEL_WIDE) s->wide = true;
    if (b->envKernel == TRM_KERNEL_QUAD && !quad_ratio_too_high(b->c)) s->wide = false;
    if (!b->c.upsample) {
        s->hist = (2u * (uint32_t)b->d.padSize + 3u) & ~3u;
        if ((rc = s->dHist.reserve(nvoices * s->hist)) || (rc = s->dTubeOff.reserve(nvoices)) || (rc = s->dTubeOff0.reserve(nvoices))) {
            delete s;
            trm_batch_destroy(b);
            return rc;
        }
    }
    if ((rc = s->dState.reserve(((nvoices + 63) / 64 * 64) * trm::kStreamFloats)) || (rc = s->dLast.reserve(nvoices * 16)) || (rc = s->dFrameOff.reserve(nvoices)) || (rc = s->dOutOff.reserve(nvoices)) ||
        (rc = s->dNFrames.reserve(nvoices)) || (rc = s->dNSamples.reserve(nvoices)) || (rc = s->dMax.reserve(nvoices))) {
        trm_stream_destroy(s);
        return rc;
    }
    // The noise sequence of the first 16 s (24 s with ensure_noise's head-room) is fetched now, not chunk by chunk: extending it is
    // a serial kernel, a synchronisation and a re-upload, i.e. a chunk that takes 2 ms longer than its neighbours (the sequence is
    // generated once per process, later streams only upload it).
    if ((rc = ensure_noise(b, 16u * (uint32_t)b->d.sampleRate, b->stream))) {
        trm_stream_destroy(s);
        return rc;
    }
    *out = s;
    return TRM_OK;
}

void trm_stream_destroy(trm_stream *s)
{
    if (!s) return;
    if (s->b) (void)hipSetDevice(s->b->device);
    trm_batch *b = s->b;
    if (s->chunkDone) (void)hipEventDestroy(s->chunkDone);
    delete s;               // device buffers first (the batch owns the stream they were used on)
    trm_batch_destroy(b);
}

// converter outputs k with read position e_k = (k * inc) >> 16 <= lastSample, i.e. k < result
static uint64_t outputs_through(uint64_t lastSamplePlusOne, uint32_t inc)
{
    if (lastSamplePlusOne == 0) return 0;
    return ((lastSamplePlusOne << 16) - 1) / inc + 1;
}

int trm_stream_set_mode(trm_stream *s, int mode)
{
    if (!s) return fail(TRM_EINVAL, "null stream");
    if (mode != TRM_STREAM_MODE_FRAMEWORK && mode != TRM_STREAM_MODE_TRACT) return fail(TRM_EINVAL, "unknown stream mode %d", mode);
    if (s->haveLast) return fail(TRM_EINVAL, "the stream's mode can only change between utterances (before the first push or after finish)");
    s->mode = mode;
    s->b->c.fricGain = mode == TRM_STREAM_MODE_TRACT ? 10.0f : 1.0f;      // Applications/TRAcT/tube.c:1371
    if (mode != TRM_STREAM_MODE_TRACT && s->b->c.controlPeriod != s->controlPeriod0) {      // (slices are TRAcT order's)
        s->mode = TRM_STREAM_MODE_TRACT;
        int rc = trm_stream_set_slice(s, 0);
        s->mode = mode;
        if (rc) return rc;
    }
    return TRM_OK;
}

int trm_stream_mode(const trm_stream *s) { return s ? s->mode : TRM_STREAM_MODE_FRAMEWORK; }

int trm_stream_set_slice(trm_stream *s, uint32_t tube_samples)
{
    if (!s) return fail(TRM_EINVAL, "null stream");
    if (s->mode != TRM_STREAM_MODE_TRACT) return fail(TRM_EINVAL, "a slice shorter than the control period needs held parameters: TRM_STREAM_MODE_TRACT");
    if (s->haveLast) return fail(TRM_EINVAL, "the slice length can only change between utterances (before the first push or after finish)");
    const uint32_t cp = tube_samples ? tube_samples : (uint32_t)s->controlPeriod0;
    if (cp < 4 || cp > 0x100000u) return fail(TRM_EINVAL, "slice of %u tube samples", tube_samples);
    // the kernels' "control period" is the run of samples one frame row stands for; nothing else of the tube depends on it
    // (the sample rate and everything derived from it were fixed when the batch was created)
    trm_batch *b = s->b;
    b->c.controlPeriod = (int32_t)cp;
    b->c.invControlPeriod = (float)(1.0 / cp);
    b->c.invControlPeriodD = 1.0 / cp;
    b->d.controlPeriod = (int32_t)cp;
    s->shapeRows = 0;              // (the chunk shapes are in frames: re-upload the index arrays)
    return TRM_OK;
}

uint32_t trm_stream_slice(const trm_stream *s) { return s ? (uint32_t)s->b->c.controlPeriod : 0u; }
int trm_stream_kernel(const trm_stream *s) { return s ? (s->wide ? TRM_KERNEL_WIDE : TRM_KERNEL_QUAD) : TRM_KERNEL_AUTO; }

size_t trm_stream_samples_for_push(const trm_stream *s, size_t nframes)
{
    if (!s || nframes == 0) return 0;
    const uint64_t periods = (s->haveLast || s->mode == TRM_STREAM_MODE_TRACT) ? nframes : nframes - 1;
    const uint64_t N = periods * (uint64_t)s->b->d.controlPeriod;
    return (size_t)(outputs_through(s->nBase + N, s->b->c.timeRegisterIncrement) - s->kBase);
}

size_t trm_stream_samples_for_finish(const trm_stream *s)
{
    if (!s || !s->haveLast) return 0;
    const uint64_t total = s->nBase + 2ull * (uint64_t)s->b->d.padSize;
    const uint32_t inc = s->b->c.timeRegisterIncrement;
    return (size_t)((total * 65536ull + inc - 1) / inc - s->kBase);
}

// One chunk on the device: control periods from the stream's last frame through the pushed frames `d_pushed` (device,
// [nvoices][nframes][16]), or the converter's flush; PCM to d_out (device, voice v at d_out + v * out_pitch).  Everything
// is work on `st`, ordered behind the chunk before it (which may have run on another stream: an event).  The host is made
// to wait only when the chunk's shape changes (the index arrays are re-uploaded) or the noise sequence has to grow.
static int stream_chunk_device_impl(trm_stream *s, const float *d_pushed, size_t nframes, bool flush, float *d_out, size_t out_pitch,
                                    uint32_t *nout, hipStream_t st);
static int stream_chunk_device(trm_stream *s, const float *d_pushed, size_t nframes, bool flush, float *d_out, size_t out_pitch,
                               uint32_t *nout, hipStream_t st)
{
    if (s->haveChunk && st != s->lastStream) HIP_TRY(hipStreamWaitEvent(st, s->chunkDone, 0));
    int rc = stream_chunk_device_impl(s, d_pushed, nframes, flush, d_out, out_pitch, nout, st);
    if (rc) return rc;
    if (!s->chunkDone) HIP_TRY(hipEventCreateWithFlags(&s->chunkDone, hipEventDisableTiming));
    HIP_TRY(hipEventRecord(s->chunkDone, st));
    s->lastStream = st;
    s->haveChunk = true;
    return TRM_OK;
}

static int stream_chunk_device_impl(trm_stream *s, const float *d_pushed, size_t nframes, bool flush, float *d_out, size_t out_pitch,
                               uint32_t *nout, hipStream_t st)
{
    trm_batch *b = s->b;
    const size_t V = s->nvoices;
    const uint32_t CP = (uint32_t)b->d.controlPeriod, inc = b->c.timeRegisterIncrement;
    const bool tract = s->mode == TRM_STREAM_MODE_TRACT;
    // TRAcT order: every frame is one control period of HELD parameters, the utterance's first one included; the kernel
    // runs period p on row p + 1 alone (stream_flags bit 2), so row 0 only has to exist
    const bool leadRow = s->haveLast || (tract && !flush);
    const size_t rows = (flush ? 0 : nframes) + (leadRow ? 1 : 0);            // frame rows per voice on the device
    if (rows == 0) { if (nout) *nout = 0; return TRM_OK; }
    const uint64_t N = (uint64_t)(rows - 1) * CP;
    const uint64_t kEnd = flush ? ((s->nBase + 2ull * (uint64_t)b->d.padSize) * 65536ull + inc - 1) / inc
                                : outputs_through(s->nBase + N, inc);
    const uint64_t count = kEnd - s->kBase;
    if (nout) *nout = (uint32_t)count;
    if (s->nBase + N + 2ull * (uint64_t)b->d.padSize + 512 > 0x7FFFFFFFull || kEnd > 0xFFFFFFFFull)
        return fail(TRM_ERANGE, "stream too long");
    if (count > 0 && (!d_out || out_pitch < count)) return fail(TRM_EINVAL, "output pitch %zu < %llu samples", out_pitch, (unsigned long long)count);
    int rc;
    const bool down = !b->c.upsample;
    const size_t rowPitch = ((size_t)s->hist + (size_t)N + 2u * (size_t)b->d.padSize + 3u) & ~(size_t)3;     // down-sampling streams
    if ((rc = s->dFrames.reserve(V * rows * 16))) return rc;
    // the rows: [lead row | pushed frames] per voice
    if (leadRow) {
        const float *src = s->haveLast ? s->dLast.p : d_pushed;
        const size_t spitch = s->haveLast ? 16 : nframes * 16;
        HIP_TRY(hipMemcpy2DAsync(s->dFrames.p, rows * 16 * sizeof(float), src, spitch * sizeof(float), 16 * sizeof(float), V, hipMemcpyDeviceToDevice, st));
    }
    if (!flush)
        HIP_TRY(hipMemcpy2DAsync(s->dFrames.p + (leadRow ? 16 : 0), rows * 16 * sizeof(float), d_pushed, nframes * 16 * sizeof(float),
                                 nframes * 16 * sizeof(float), V, hipMemcpyDeviceToDevice, st));
    // the index arrays depend on the chunk's shape only: uploaded when it changes (host copies live in the stream object)
    if (s->shapeRows != rows || s->shapePitch != out_pitch || (down && s->shapeRowPitch != rowPitch)) {
        s->hFrameOff.resize(V); s->hOutOff.resize(V); s->hNFrames.assign(V, (uint32_t)rows);
        for (size_t v = 0; v < V; v++) { s->hFrameOff[v] = v * rows; s->hOutOff[v] = v * out_pitch; }
        HIP_TRY(hipStreamSynchronize(st));          // (an earlier launch may still be reading the arrays)
        HIP_TRY(hipMemcpyAsync(s->dFrameOff.p, s->hFrameOff.data(), V * sizeof(uint64_t), hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync(s->dOutOff.p, s->hOutOff.data(), V * sizeof(uint64_t), hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync(s->dNFrames.p, s->hNFrames.data(), V * sizeof(uint32_t), hipMemcpyHostToDevice, st));
        if (down) {
            s->hTubeOff0.resize(V); s->hTubeOff.resize(V);
            for (size_t v = 0; v < V; v++) { s->hTubeOff0[v] = v * rowPitch; s->hTubeOff[v] = v * rowPitch + s->hist; }
            HIP_TRY(hipMemcpyAsync(s->dTubeOff0.p, s->hTubeOff0.data(), V * sizeof(uint64_t), hipMemcpyHostToDevice, st));
            HIP_TRY(hipMemcpyAsync(s->dTubeOff.p, s->hTubeOff.data(), V * sizeof(uint64_t), hipMemcpyHostToDevice, st));
        }
        HIP_TRY(hipStreamSynchronize(st));          // (pageable sources: the copies are done before the vectors can change)
        s->shapeRows = rows; s->shapePitch = out_pitch; s->shapeRowPitch = rowPitch;
    }
    if ((rc = ensure_noise(b, (uint32_t)(s->nBase + N) + 2u * (uint32_t)b->d.padSize + 256u, st))) return rc;
    if (N > 0 || flush) {
        trm::TubeArgs a;
        a.frames = s->dFrames.p;
        a.frame_offset = s->dFrameOff.p;
        a.nframes = s->dNFrames.p;
        a.out = d_out;
        a.out_offset = s->dOutOff.p;
        a.number_samples = s->dNSamples.p;
        a.max_sample = s->dMax.p;
        a.lp_noise = b->dNoise.p + s->nBase;         // the voice-independent noise sequence continues where it stopped
        a.src_rows = b->dRows;
        a.sine = b->dSine;
        a.tube_out = nullptr;
        a.tube_offset = nullptr;
        if (down) {
            // rows of [history | the chunk's tube samples (| the flush zeros)]; the tube stage writes behind the history
            if ((rc = s->dTube.reserve(V * rowPitch + 4))) return rc;
            if (s->first) HIP_TRY(hipMemsetAsync(s->dHist.p, 0, V * s->hist * sizeof(float), st));
            HIP_TRY(hipMemcpy2DAsync(s->dTube.p, rowPitch * sizeof(float), s->dHist.p, s->hist * sizeof(float), s->hist * sizeof(float), V,
                                     hipMemcpyDeviceToDevice, st));
            a.tube_out = s->dTube.p;
            a.tube_offset = s->dTubeOff.p;
        }
        a.nvoices = (uint32_t)V;
        a.max_nframes = 0xFFFFFFFFu;          // (nframes is this function's own array)
        a.stamps = nullptr;
        a.stream_state = s->dState.p;
        a.stream_flags = (s->first ? 1u : 0u) | (flush ? 2u : 0u) | (tract ? 4u : 0u);
        a.stream_n_base = (uint32_t)s->nBase;
        a.stream_k_base = (uint32_t)s->kBase;
        a.stream_k_end = (uint32_t)kEnd;
        if (s->wide) HIP_TRY(trm::launch_tube(b->c, a, st));
        else HIP_TRY(trm::launch_tube_quad(b->c, a, st, b->cus));
        b->lastKernel = s->wide ? TRM_KERNEL_WIDE : TRM_KERNEL_QUAD;
        s->first = false;
        if (down) {
            if (count > 0) {
                trm::DownArgs d;
                d.tube = s->dTube.p;
                d.tube_offset = s->dTubeOff0.p;
                d.nframes = s->dNFrames.p;
                d.out = d_out;
                d.out_offset = s->dOutOff.p;
                d.number_samples = s->dNSamples.p;
                d.max_sample = s->dMax.p;
                d.fine = b->dFine;
                d.nvoices = (uint32_t)V;
                d.max_nframes = 0xFFFFFFFFu;
                d.rows = b->dDownRows;
                d.lmax = b->downL; d.rmax = b->downR; d.pitch = b->downPitch;
                d.stream = 1;
                d.n_origin = (long long)s->nBase - (long long)s->hist;
                d.n_hi = (long long)(s->nBase + N + (flush ? 2ull * (uint64_t)b->d.padSize : 0ull));
                d.k_base = (uint32_t)s->kBase;
                d.k_end = (uint32_t)kEnd;
                HIP_TRY(trm::launch_downsample(b->c, d, st));
            } else {
                HIP_TRY(hipMemsetAsync(s->dMax.p, 0, V * sizeof(float), st));
            }
            // the next chunk's history: the last `hist` tube samples so far (row positions N .. N + hist - 1)
            HIP_TRY(hipMemcpy2DAsync(s->dHist.p, s->hist * sizeof(float), s->dTube.p + N, rowPitch * sizeof(float), s->hist * sizeof(float), V,
                                     hipMemcpyDeviceToDevice, st));
        }
        if (tract && count > 0) {
            // tube.c:1177 multiplies the tube-rate sample by 100 before its converter; the converter is linear, so the gain
            // is applied to what it returns (one fp32 rounding of difference)
            HIP_TRY(trm::launch_gain(d_out, out_pitch, (uint32_t)count, (uint32_t)V, s->dMax.p, 100.0f, st));
        }
    } else {
        HIP_TRY(hipMemsetAsync(s->dMax.p, 0, V * sizeof(float), st));
    }
    if (!flush) {
        // the frame the next control period starts from
        HIP_TRY(hipMemcpy2DAsync(s->dLast.p, 16 * sizeof(float), d_pushed + (nframes - 1) * 16, nframes * 16 * sizeof(float), 16 * sizeof(float), V,
                                 hipMemcpyDeviceToDevice, st));
    }
    s->nBase += N;
    s->kBase = kEnd;
    return TRM_OK;
}

static void stream_after_push(trm_stream *s) { s->haveLast = true; }
static void stream_after_finish(trm_stream *s)
{
    s->haveLast = false;          // the next push opens a new utterance: tube at rest, converter pre-roll
    s->first = true;
    s->nBase = 0;
    s->kBase = 0;
}

// host-buffer form: H2D of the frames, the chunk, D2H of its PCM
static int stream_chunk(trm_stream *s, const float *frames, size_t nframes, bool flush, float *out, size_t out_pitch,
                        uint32_t *nout, float *max_out)
{
    trm_batch *b = s->b;
    const size_t V = s->nvoices;
    HIP_TRY(hipSetDevice(b->device));
    hipStream_t st = b->stream;
    int rc;
    const size_t count = flush ? trm_stream_samples_for_finish(s) : trm_stream_samples_for_push(s, nframes);
    if (count > 0 && (!out || out_pitch < count)) return fail(TRM_EINVAL, "output pitch %zu < %zu samples", out_pitch, count);
    if (!flush) {
        if ((rc = s->dPushed.reserve(V * nframes * 16))) return rc;
        HIP_TRY(hipMemcpyAsync(s->dPushed.p, frames, V * nframes * 16 * sizeof(float), hipMemcpyHostToDevice, st));
    }
    if ((rc = s->dOut.reserve(V * count + 64))) return rc;
    uint32_t got = 0;
    if ((rc = stream_chunk_device(s, flush ? nullptr : s->dPushed.p, nframes, flush, s->dOut.p, count, &got, st))) return rc;
    if (nout) *nout = got;
    if (got > 0) {
        s->hostOut.resize(V * (size_t)got);
        HIP_TRY(hipMemcpyAsync(s->hostOut.data(), s->dOut.p, V * (size_t)got * sizeof(float), hipMemcpyDeviceToHost, st));
    }
    std::vector<float> mx(V, 0.0f);
    HIP_TRY(hipMemcpyAsync(mx.data(), s->dMax.p, V * sizeof(float), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    for (size_t v = 0; v < V && got > 0; v++) memcpy(out + v * out_pitch, &s->hostOut[v * (size_t)got], (size_t)got * sizeof(float));
    if (max_out) memcpy(max_out, mx.data(), V * sizeof(float));
    return TRM_OK;
}

int trm_stream_push(trm_stream *s, const float *frames, size_t nframes, float *out, size_t out_pitch, uint32_t *nout, float *max_out)
{
    if (!s || !frames || nframes == 0) return fail(TRM_EINVAL, "null argument / no frames");
    int rc = stream_chunk(s, frames, nframes, false, out, out_pitch, nout, max_out);
    if (rc) return rc;
    stream_after_push(s);
    return TRM_OK;
}

int trm_stream_finish(trm_stream *s, float *out, size_t out_pitch, uint32_t *nout, float *max_out)
{
    if (!s) return fail(TRM_EINVAL, "null stream");
    if (!s->haveLast) { if (nout) *nout = 0; return TRM_OK; }
    int rc = stream_chunk(s, nullptr, 0, true, out, out_pitch, nout, max_out);
    if (rc) return rc;
    stream_after_finish(s);
    return TRM_OK;
}

int trm_stream_push_device(trm_stream *s, const float *d_frames, size_t nframes, float *d_out, size_t out_pitch, uint32_t *nout,
                           float *d_max_out, void *stream)
{
    if (!s || !d_frames || nframes == 0) return fail(TRM_EINVAL, "null argument / no frames");
    HIP_TRY(hipSetDevice(s->b->device));
    hipStream_t st = (hipStream_t)stream;
    int rc = stream_chunk_device(s, d_frames, nframes, false, d_out, out_pitch, nout, st);
    if (rc) return rc;
    if (d_max_out) HIP_TRY(hipMemcpyAsync(d_max_out, s->dMax.p, s->nvoices * sizeof(float), hipMemcpyDeviceToDevice, st));
    stream_after_push(s);
    return TRM_OK;
}

int trm_stream_finish_device(trm_stream *s, float *d_out, size_t out_pitch, uint32_t *nout, float *d_max_out, void *stream)
{
    if (!s) return fail(TRM_EINVAL, "null stream");
    if (!s->haveLast) { if (nout) *nout = 0; return TRM_OK; }
    HIP_TRY(hipSetDevice(s->b->device));
    hipStream_t st = (hipStream_t)stream;
    int rc = stream_chunk_device(s, nullptr, 0, true, d_out, out_pitch, nout, st);
    if (rc) return rc;
    if (d_max_out) HIP_TRY(hipMemcpyAsync(d_max_out, s->dMax.p, s->nvoices * sizeof(float), hipMemcpyDeviceToDevice, st));
    stream_after_finish(s);
    return TRM_OK;
}

int trm_batch_set_kernel(trm_batch *b, int kernel)
{
    if (!b) return fail(TRM_EINVAL, "null batch");
    if (kernel != TRM_KERNEL_AUTO && kernel != TRM_KERNEL_WIDE && kernel != TRM_KERNEL_QUAD && kernel != TRM_KERNEL_OCT) return fail(TRM_EINVAL, "unknown kernel form %d", kernel);
    b->kernel = kernel;
    return TRM_OK;
}

int trm_batch_last_kernel(const trm_batch *b) { return b ? b->lastKernel : TRM_KERNEL_AUTO; }

int trm_batch_set_timing(trm_batch *b, int on)
{
    if (!b) return fail(TRM_EINVAL, "null batch");
    b->timing = on != 0;
    return TRM_OK;
}

int trm_batch_kernel_time_ms(trm_batch *b, double *total_ms, uint32_t *launches)
{
    if (!b || !total_ms || !launches) return fail(TRM_EINVAL, "null argument");
    int rc = fold_events(b, true);
    if (rc) return rc;
    const double sum = b->timedMs;
    const uint32_t n = b->timedLaunches;
    b->timedMs = 0.0;
    b->timedLaunches = 0;
    *total_ms = sum;
    *launches = n;
    return TRM_OK;
}

#ifdef TRM_STAMP
extern "C" int trm_debug_stamps(unsigned long long *host_out, size_t n)
{
    if (!g_stampPtr) return TRM_EINVAL;
    if (hipDeviceSynchronize() != hipSuccess) return TRM_EHIP;
    return hipMemcpy(host_out, g_stampPtr, n * sizeof(unsigned long long), hipMemcpyDeviceToHost) == hipSuccess ? TRM_OK : TRM_EHIP;
}
#endif

int trm_batch_noise_table(trm_batch *b, float *host_out, size_t n)
{
    if (!b || !host_out) return fail(TRM_EINVAL, "null argument");
    if (n > 0x7FFFFFF0ull) return fail(TRM_ERANGE, "too long");
    HIP_TRY(hipSetDevice(b->device));
    int rc = ensure_noise(b, (uint32_t)n, b->stream);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(host_out, b->dNoise.p, n * sizeof(float), hipMemcpyDeviceToHost, b->stream));
    HIP_TRY(hipStreamSynchronize(b->stream));
    return TRM_OK;
}

// host-buffer entry: fp32 PCM (out) or the containers' int16 (out16, mono or interleaved stereo), not both
static int synthesize_host_impl(trm_batch *b, size_t nvoices, const float *frames, const uint64_t *frame_offset,
                                const uint32_t *nframes, float *out, int16_t *out16, int for_wav_data,
                                const uint64_t *out_offset, uint32_t *number_samples, float *max_sample)
{
    if (!b) return fail(TRM_EINVAL, "null batch");
    if (nvoices == 0) return TRM_OK;
    if (!frames || !frame_offset || !nframes || (!out && !out16) || !out_offset || !number_samples || !max_sample)
        return fail(TRM_EINVAL, "null pointer");
    HIP_TRY(hipSetDevice(b->device));
    uint64_t frameRows = 0, outEnd = 0, outBegin = ~0ull, outSum = 0;
    uint32_t maxFrames = 0;
    for (size_t v = 0; v < nvoices; v++) {
        uint64_t fe = frame_offset[v] + nframes[v];
        if (fe > frameRows) frameRows = fe;
        const uint64_t ns = trm_batch_samples_for_frames(b, nframes[v]);
        uint64_t oe = out_offset[v] + ns;
        if (oe > outEnd) outEnd = oe;
        if (ns > 0 && out_offset[v] < outBegin) outBegin = out_offset[v];
        outSum += ns;
        if (nframes[v] > maxFrames) maxFrames = nframes[v];
    }
    if (outBegin > outEnd) outBegin = outEnd;
    // The header promises writes at out + out_offset[v] only.  Dense spans (the usual case: the voices tile
    // [outBegin, outEnd) exactly) come back in one copy; a span with gaps is copied voice by voice, so that what
    // lies between the voices in the caller's buffer is left alone.
    const bool dense = outSum == outEnd - outBegin;
    if (frameRows == 0) frameRows = 1;
    const size_t ch = b->params.channels == 2 ? 2 : 1;
    int rc;
    if ((rc = b->dFrames.reserve(frameRows * 16)) || (rc = b->dOut.reserve(outEnd + 1)) ||
        (rc = b->dFrameOff.reserve(nvoices)) || (rc = b->dOutOff.reserve(nvoices)) ||
        (rc = b->dNFrames.reserve(nvoices)) || (rc = b->dNSamples.reserve(nvoices)) || (rc = b->dMax.reserve(nvoices)))
        return rc;
    if (out16 && (rc = b->dOut16.reserve(outEnd * ch + 1))) return rc;
    hipStream_t s = b->stream;
    // Ragged batches: the kernels' voice index is put in order of decreasing length, so that the voices of a workgroup
    // (16 or 64 consecutive indices) end together instead of every workgroup lasting as long as the longest voice of
    // the batch: a workgroup that ends early frees its CU for the next one.  Only the three index arrays are permuted
    // (frames and PCM stay where the caller's offsets put them); the per-voice results are put back in caller order.
    std::vector<uint32_t> perm;
    std::vector<uint64_t> pFrameOff, pOutOff;
    std::vector<uint32_t> pNFrames, pNs;
    std::vector<float> pMx;
    {
        bool sorted = true;
        for (size_t v = 1; v < nvoices && sorted; v++) sorted = nframes[v] <= nframes[v - 1];
        if (!sorted && nvoices > 16) {
            perm.resize(nvoices);
            for (size_t v = 0; v < nvoices; v++) perm[v] = (uint32_t)v;
            std::stable_sort(perm.begin(), perm.end(), [&](uint32_t x, uint32_t y) { return nframes[x] > nframes[y]; });
            pFrameOff.resize(nvoices); pOutOff.resize(nvoices); pNFrames.resize(nvoices); pNs.resize(nvoices); pMx.resize(nvoices);
            for (size_t i = 0; i < nvoices; i++) {
                pFrameOff[i] = frame_offset[perm[i]];
                pOutOff[i] = out_offset[perm[i]];
                pNFrames[i] = nframes[perm[i]];
            }
        }
    }
    const bool permuted = !perm.empty();
    HIP_TRY(hipMemcpyAsync(b->dFrames.p, frames, frameRows * 16 * sizeof(float), hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(b->dFrameOff.p, permuted ? pFrameOff.data() : frame_offset, nvoices * sizeof(uint64_t), hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(b->dOutOff.p, permuted ? pOutOff.data() : out_offset, nvoices * sizeof(uint64_t), hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(b->dNFrames.p, permuted ? pNFrames.data() : nframes, nvoices * sizeof(uint32_t), hipMemcpyHostToDevice, s));
    {
        // what the time-split planner may know about a ragged batch: its control periods in all (the kernels' voice index
        // runs from the longest voice down: the segments' workgroups fill up front to back)
        uint64_t tp = 0;
        for (size_t v = 0; v < nvoices; v++) tp += nframes[v] > 1 ? nframes[v] - 1 : 0;
        bool desc = true;
        for (size_t v = 1; v < nvoices && desc && !permuted; v++) desc = nframes[v] <= nframes[v - 1];
        b->hintTotalPeriods = (permuted || desc) ? tp : 0;
        const uint32_t *order = permuted ? pNFrames.data() : nframes;
        b->hintFrames.assign(order, order + nvoices);
    }
    rc = trm_batch_synthesize_device(b, nvoices, b->dFrames.p, b->dFrameOff.p, b->dNFrames.p, maxFrames, b->dOut.p,
                                     b->dOutOff.p, b->dNSamples.p, b->dMax.p, s);
    if (rc) return rc;
    if (out16) {
        rc = trm_batch_scale_to_int16_device(b, nvoices, b->dOut.p, b->dOutOff.p, b->dNSamples.p, b->dMax.p, b->dOut16.p,
                                             for_wav_data, s);
        if (rc) return rc;
        if (dense && outEnd > outBegin)
            HIP_TRY(hipMemcpyAsync(out16 + outBegin * ch, b->dOut16.p + outBegin * ch, (outEnd - outBegin) * ch * sizeof(int16_t), hipMemcpyDeviceToHost, s));
        for (size_t v = 0; !dense && v < nvoices; v++)
            if (const uint64_t ns = trm_batch_samples_for_frames(b, nframes[v]))
                HIP_TRY(hipMemcpyAsync(out16 + out_offset[v] * ch, b->dOut16.p + out_offset[v] * ch, ns * ch * sizeof(int16_t), hipMemcpyDeviceToHost, s));
    } else {
        if (dense && outEnd > outBegin)
            HIP_TRY(hipMemcpyAsync(out + outBegin, b->dOut.p + outBegin, (outEnd - outBegin) * sizeof(float), hipMemcpyDeviceToHost, s));
        for (size_t v = 0; !dense && v < nvoices; v++)
            if (const uint64_t ns = trm_batch_samples_for_frames(b, nframes[v]))
                HIP_TRY(hipMemcpyAsync(out + out_offset[v], b->dOut.p + out_offset[v], ns * sizeof(float), hipMemcpyDeviceToHost, s));
    }
    HIP_TRY(hipMemcpyAsync(permuted ? pNs.data() : number_samples, b->dNSamples.p, nvoices * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(permuted ? pMx.data() : max_sample, b->dMax.p, nvoices * sizeof(float), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    for (size_t i = 0; permuted && i < nvoices; i++) {
        number_samples[perm[i]] = pNs[i];
        max_sample[perm[i]] = pMx[i];
    }
    return TRM_OK;
}

int trm_batch_synthesize_host(trm_batch *b, size_t nvoices, const float *frames, const uint64_t *frame_offset,
                              const uint32_t *nframes, float *out, const uint64_t *out_offset,
                              uint32_t *number_samples, float *max_sample)
{
    if (!out) return fail(TRM_EINVAL, "null pointer");
    return synthesize_host_impl(b, nvoices, frames, frame_offset, nframes, out, nullptr, 0, out_offset, number_samples, max_sample);
}

int trm_batch_synthesize_host_int16(trm_batch *b, size_t nvoices, const float *frames, const uint64_t *frame_offset,
                                    const uint32_t *nframes, int16_t *out16, const uint64_t *out_offset,
                                    uint32_t *number_samples, float *max_sample, int for_wav_data)
{
    if (!out16) return fail(TRM_EINVAL, "null pointer");
    return synthesize_host_impl(b, nvoices, frames, frame_offset, nframes, nullptr, out16, for_wav_data, out_offset, number_samples,
                                max_sample);
}

// ------------------------------------------------------------------ several devices from one process
int trm_shard_voices(const uint32_t *nframes, size_t nvoices, size_t nshards, size_t *bounds)
{
    if (!bounds || nshards == 0 || (nvoices && !nframes)) return fail(TRM_EINVAL, "null argument / no shards");
    // cost of a voice = its frames + 1 (a launch lasts as long as its longest voice, but the bytes moved and the
    // lanes occupied go with the sum); boundary g sits where the running cost passes g/nshards of the total
    uint64_t total = 0;
    for (size_t v = 0; v < nvoices; v++) total += (uint64_t)nframes[v] + 1u;
    bounds[0] = 0;
    size_t v = 0;
    uint64_t run = 0;
    for (size_t g = 1; g < nshards; g++) {
        const uint64_t target = (total * g + nshards / 2) / nshards;
        while (v < nvoices && run + ((uint64_t)nframes[v] + 1u) / 2 < target) run += (uint64_t)nframes[v++] + 1u;
        bounds[g] = v;
    }
    bounds[nshards] = nvoices;
    return TRM_OK;
}

struct trm_multi {
    std::vector<trm_batch *> b;
};

int trm_multi_create(const trm_input_params *params, const int *devices, size_t ndevices, trm_multi **out)
{
    if (!params || !devices || !out || ndevices == 0) return fail(TRM_EINVAL, "null argument / no devices");
    *out = nullptr;
    trm_multi *m = new trm_multi;
    for (size_t g = 0; g < ndevices; g++) {
        trm_batch *b = nullptr;
        int rc = trm_batch_create(params, devices[g], &b);
        if (rc) {
            trm_multi_destroy(m);
            return rc;
        }
        m->b.push_back(b);
    }
    *out = m;
    return TRM_OK;
}

void trm_multi_destroy(trm_multi *m)
{
    if (!m) return;
    for (trm_batch *b : m->b) trm_batch_destroy(b);
    delete m;
}

static int multi_synthesize_impl(trm_multi *m, size_t nvoices, const float *frames, const uint64_t *frame_offset,
                                 const uint32_t *nframes, float *out, int16_t *out16, int for_wav_data, const uint64_t *out_offset,
                                 uint32_t *number_samples, float *max_sample)
{
    if (!m) return fail(TRM_EINVAL, "null handle");
    if (nvoices == 0) return TRM_OK;
    if (!frames || !frame_offset || !nframes || (!out && !out16) || !out_offset || !number_samples || !max_sample)
        return fail(TRM_EINVAL, "null pointer");
    const size_t G = m->b.size();
    std::vector<size_t> bounds(G + 1);
    trm_shard_voices(nframes, nvoices, G, bounds.data());
    struct Shard {
        size_t lo, hi;
        uint64_t fLo, oLo, oHi;
        std::vector<uint64_t> foff, ooff;
        int rc = TRM_OK;
        std::string err;
    };
    std::vector<Shard> sh(G);
    for (size_t g = 0; g < G; g++) {
        Shard &s = sh[g];
        s.lo = bounds[g]; s.hi = bounds[g + 1];
        s.fLo = s.oLo = ~0ull; s.oHi = 0;
        for (size_t v = s.lo; v < s.hi; v++) {
            if (nframes[v] && frame_offset[v] < s.fLo) s.fLo = frame_offset[v];
            const uint64_t n = trm_batch_samples_for_frames(m->b[g], nframes[v]);
            if (n && out_offset[v] < s.oLo) s.oLo = out_offset[v];
            if (n && out_offset[v] + n > s.oHi) s.oHi = out_offset[v] + n;
        }
        if (s.fLo == ~0ull) s.fLo = 0;
        if (s.oLo == ~0ull) s.oLo = s.oHi = 0;
        for (size_t v = s.lo; v < s.hi; v++) {
            s.foff.push_back(nframes[v] ? frame_offset[v] - s.fLo : 0);
            s.ooff.push_back(trm_batch_samples_for_frames(m->b[g], nframes[v]) ? out_offset[v] - s.oLo : 0);
        }
    }
    // a device copies back the whole span [oLo, oHi) of its shard: spans of different shards must be disjoint
    for (size_t g = 0; g < G; g++)
        for (size_t h = g + 1; h < G; h++)
            if (sh[g].oHi > sh[g].oLo && sh[h].oHi > sh[h].oLo && sh[g].oLo < sh[h].oHi && sh[h].oLo < sh[g].oHi)
                return fail(TRM_EINVAL, "output ranges of shards %zu and %zu interleave", g, h);
    std::vector<std::thread> th;
    for (size_t g = 0; g < G; g++) {
        if (sh[g].hi == sh[g].lo) continue;
        th.emplace_back([&, g]() {
            Shard &s = sh[g];
            const size_t ch = m->b[g]->params.channels == 2 ? 2 : 1;
            s.rc = synthesize_host_impl(m->b[g], s.hi - s.lo, frames + s.fLo * 16, s.foff.data(), nframes + s.lo,
                                        out ? out + s.oLo : nullptr, out16 ? out16 + s.oLo * ch : nullptr, for_wav_data, s.ooff.data(),
                                        number_samples + s.lo, max_sample + s.lo);
            if (s.rc) s.err = trm_last_error();      // (the detail text is thread-local: carry it to the caller's thread)
        });
    }
    for (std::thread &t : th) t.join();
    for (size_t g = 0; g < G; g++)
        if (sh[g].rc) return fail(sh[g].rc, "shard %zu (device %d): %s", g, m->b[g]->device, sh[g].err.c_str());
    return TRM_OK;
}

int trm_multi_synthesize_host(trm_multi *m, size_t nvoices, const float *frames, const uint64_t *frame_offset,
                              const uint32_t *nframes, float *out, const uint64_t *out_offset,
                              uint32_t *number_samples, float *max_sample)
{
    if (!out) return fail(TRM_EINVAL, "null pointer");
    return multi_synthesize_impl(m, nvoices, frames, frame_offset, nframes, out, nullptr, 0, out_offset, number_samples, max_sample);
}

int trm_multi_synthesize_host_int16(trm_multi *m, size_t nvoices, const float *frames, const uint64_t *frame_offset,
                                    const uint32_t *nframes, int16_t *out16, const uint64_t *out_offset,
                                    uint32_t *number_samples, float *max_sample, int for_wav_data)
{
    if (!out16) return fail(TRM_EINVAL, "null pointer");
    return multi_synthesize_impl(m, nvoices, frames, frame_offset, nframes, nullptr, out16, for_wav_data, out_offset, number_samples,
                                 max_sample);
}

// ------------------------------------------------------------------ control-track generation
float trm_drift_seed_after(float seed, size_t ngenerated)
{
    // MMDriftGenerator.m:65-70 in float, one rounding per operation (x86-64 semantics, like the kernel and the oracle)
    volatile float sd = seed != 0.0f ? seed : 0.7892347f;
    for (size_t i = 0; i < ngenerated; i++) {
        volatile float temp = sd * 377.0f;
        sd = temp - (float)(int32_t)temp;
    }
    return sd;
}

int trm_events_count_frames(const uint32_t *times, size_t n, const trm_intonation *s, size_t *nframes)
{
    if (!s || !nframes || (n && !times)) return fail(TRM_EINVAL, "null argument");
    *nframes = 0;
    if (n < 2) return TRM_OK;
    uint64_t start = s->startTime_ms, end = s->endTime_ms;
    if (start == 0 && end == 0) end = ~0ull;                      // EventList.m:892-894
    size_t i = 1, count = 0;
    uint64_t t = 0, nextTime = times[1];
    while (i < n) {                                               // the time stepping of EventList.m:970-1027
        if (t >= start && t <= end) count++;
        t += 4;
        if (t >= nextTime) {
            i++;
            if (i == n) break;
            nextTime = times[i];
        }
    }
    *nframes = count;
    return TRM_OK;
}

int trm_batch_generate_frames_device(trm_batch *b, size_t nvoices, const uint32_t *d_event_times, const double *d_event_values,
                                     const uint64_t *d_event_offset, const uint32_t *d_nevents, const trm_intonation *settings,
                                     float *d_frames, const uint64_t *d_frame_offset, uint32_t *d_nframes_out, void *stream_)
{
    if (!b) return fail(TRM_EINVAL, "null batch");
    if (nvoices == 0) return TRM_OK;
    if (!d_event_times || !d_event_values || !d_event_offset || !d_nevents || !settings || !d_frames || !d_frame_offset || !d_nframes_out)
        return fail(TRM_EINVAL, "null pointer");
    if (nvoices > 0x7FFFFFFFull) return fail(TRM_EINVAL, "too many voices");
    HIP_TRY(hipSetDevice(b->device));
    trm::TrackArgs a;
    a.event_times = d_event_times;
    a.event_values = d_event_values;
    a.event_offset = d_event_offset;
    a.nevents = d_nevents;
    a.frames = d_frames;
    a.frame_offset = d_frame_offset;
    a.nframes_out = d_nframes_out;
    a.settings = *settings;
    a.nvoices = (uint32_t)nvoices;
    HIP_TRY(trm::launch_tracks(a, (hipStream_t)stream_));
    return TRM_OK;
}

int trm_batch_generate_frames_host(trm_batch *b, const uint32_t *times, const double *values, size_t nevents,
                                   const trm_intonation *settings, float *frames_out, size_t frames_cap, size_t *nframes)
{
    if (!b || !settings || !nframes || (nevents && (!times || !values))) return fail(TRM_EINVAL, "null argument");
    size_t want = 0;
    int rc = trm_events_count_frames(times, nevents, settings, &want);
    if (rc) return rc;
    *nframes = want;
    if (want == 0) return TRM_OK;
    if (!frames_out || frames_cap < want) return fail(TRM_EINVAL, "frame buffer holds %zu rows, %zu needed", frames_cap, want);
    HIP_TRY(hipSetDevice(b->device));
    // staging buffers of this entry live in the batch object (grow-only): no allocation per utterance
    DevBuf<uint32_t> &dT = b->evT, &dN = b->evN;
    DevBuf<double> &dV = b->evV;
    DevBuf<uint64_t> &dOff = b->evOff;
    DevBuf<float> &dF = b->evF;
    if ((rc = dT.reserve(nevents)) || (rc = dV.reserve(nevents * TRM_EVENT_VALUES)) || (rc = dOff.reserve(2)) || (rc = dN.reserve(2)) ||
        (rc = dF.reserve(want * 16)))
        return rc;
    const uint64_t zero2[2] = {0, 0};
    const uint32_t ne = (uint32_t)nevents;
    hipStream_t st = b->stream;
    HIP_TRY(hipMemcpyAsync(dT.p, times, nevents * sizeof(uint32_t), hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(dV.p, values, nevents * TRM_EVENT_VALUES * sizeof(double), hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(dOff.p, zero2, sizeof zero2, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(dN.p, &ne, sizeof ne, hipMemcpyHostToDevice, st));
    rc = trm_batch_generate_frames_device(b, 1, dT.p, dV.p, dOff.p, dN.p, settings, dF.p, dOff.p + 1, dN.p + 1, st);
    if (rc) return rc;
    uint32_t got = 0;
    HIP_TRY(hipMemcpyAsync(&got, dN.p + 1, sizeof got, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(frames_out, dF.p, want * 16 * sizeof(float), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    if (got != want) return fail(TRM_EHIP, "generator wrote %u frames, %zu expected", got, want);
    return TRM_OK;
}

int trm_batch_scale_to_int16_device(trm_batch *b, size_t nvoices, const float *d_pcm, const uint64_t *d_out_offset,
                                    const uint32_t *d_number_samples, const float *d_max_sample, int16_t *d_int16,
                                    int for_wav_data, void *stream_)
{
    if (!b) return fail(TRM_EINVAL, "null batch");
    if (nvoices == 0) return TRM_OK;
    if (!d_pcm || !d_out_offset || !d_number_samples || !d_max_sample || !d_int16) return fail(TRM_EINVAL, "null device pointer");
    HIP_TRY(hipSetDevice(b->device));
    trm::ScaleArgs s;
    s.pcm = d_pcm;
    s.out_offset = d_out_offset;
    s.number_samples = d_number_samples;
    s.max_sample = d_max_sample;
    s.pcm16 = d_int16;
    s.volumeAmp = trm::io_amplitude(b->params.volume);
    s.balance = b->params.balance;
    s.channels = b->params.channels;
    s.forWavData = for_wav_data != 0;
    HIP_TRY(trm::launch_int16(s, (uint32_t)nvoices, (hipStream_t)stream_));
    return TRM_OK;
}

size_t trm_sound_file_size(const trm_input_params *params, size_t nsamples)
{
    if (!params) return 0;
    uint8_t hdr[56];
    const size_t h = trm::io_sound_file_header(*params, nsamples, hdr);
    return h ? h + nsamples * (params->channels == 2 ? 2 : 1) * 2 : 0;
}

int trm_batch_sound_files_device(trm_batch *b, size_t nvoices, const float *d_pcm, const uint64_t *d_out_offset,
                                 const uint32_t *d_number_samples, const float *d_max_sample, uint8_t *d_files,
                                 const uint64_t *d_file_offset, void *stream_)
{
    if (!b) return fail(TRM_EINVAL, "null batch");
    if (nvoices == 0) return TRM_OK;
    if (!d_pcm || !d_out_offset || !d_number_samples || !d_max_sample || !d_files || !d_file_offset) return fail(TRM_EINVAL, "null device pointer");
    HIP_TRY(hipSetDevice(b->device));
    trm::FileArgs f;
    f.s.pcm = d_pcm;
    f.s.out_offset = d_out_offset;
    f.s.number_samples = d_number_samples;
    f.s.max_sample = d_max_sample;
    f.s.pcm16 = nullptr;
    f.s.volumeAmp = trm::io_amplitude(b->params.volume);
    f.s.balance = b->params.balance;
    f.s.channels = b->params.channels;
    f.s.forWavData = 0;
    f.files = d_files;
    f.file_offset = d_file_offset;
    f.format = b->params.outputFileFormat;
    memset(f.header, 0, sizeof f.header);
    if (trm::io_sound_file_header(b->params, 0, f.header) == 0) return fail(TRM_EINVAL, "unknown sound file format %d", (int)b->params.outputFileFormat);
    HIP_TRY(trm::launch_file_images(f, (uint32_t)nvoices, (hipStream_t)stream_));
    return TRM_OK;
}

// ------------------------------------------------------------------ TRMTubeModel
int trm_tube_create(const trm_input_params *params, int device, trm_tube **out)
{
    if (!params || !out) return fail(TRM_EINVAL, "null argument");
    *out = nullptr;
    trm_batch *b = nullptr;
    int rc = trm_batch_create(params, device, &b);
    if (rc) return rc;
    trm_tube *t = new (std::nothrow) trm_tube();
    if (!t) { trm_batch_destroy(b); return fail(TRM_ENOMEM, "trm_tube"); }
    t->b = b;
    *out = t;
    return TRM_OK;
}

void trm_tube_destroy(trm_tube *t)
{
    if (!t) return;
    trm_batch_destroy(t->b);
    delete t;
}

int trm_tube_derived(const trm_tube *t, trm_derived *out)
{
    if (!t) return fail(TRM_EINVAL, "null tube");
    return trm_batch_derived(t->b, out);
}

int trm_tube_synthesize(trm_tube *t, const trm_parameters *frames, size_t nframes)
{
    if (!t || (nframes && !frames)) return fail(TRM_EINVAL, "null argument");
    t->samples.clear();
    t->numberSamples = 0;
    t->maxSample = 0.f;
    if (nframes == 0) return TRM_OK;                                    // TRMTubeModel.m:274-277
    if (nframes > 0xFFFFFFFFull) return fail(TRM_EINVAL, "too many frames");
    std::vector<float> f32(nframes * 16);
    const double *src = reinterpret_cast<const double *>(frames);
    for (size_t i = 0; i < nframes * 16; i++) f32[i] = (float)src[i];
    size_t nout = trm_batch_samples_for_frames(t->b, nframes);
    t->samples.assign(nout, 0.f);
    uint64_t foff = 0, ooff = 0;
    uint32_t nf = (uint32_t)nframes, ns = 0;
    float mx = 0.f;
    int rc = trm_batch_synthesize_host(t->b, 1, f32.data(), &foff, &nf, t->samples.data(), &ooff, &ns, &mx);
    if (rc) { t->samples.clear(); return rc; }
    t->numberSamples = ns;
    t->maxSample = mx;
    return TRM_OK;
}

size_t trm_tube_number_samples(const trm_tube *t) { return t ? t->numberSamples : 0; }
double trm_tube_maximum_sample_value(const trm_tube *t) { return t ? (double)t->maxSample : 0.0; }
const float *trm_tube_samples(const trm_tube *t) { return (t && !t->samples.empty()) ? t->samples.data() : nullptr; }

int trm_tube_print_input_data(const trm_tube *t, const trm_parameters *frames, size_t nframes)
{
    if (!t || (nframes && !frames)) return fail(TRM_EINVAL, "null argument");
    const trm_input_params &p = t->b->params;
    // -[TRMDataList printInputParameters], TRMDataList.m:251-283
    static const char *const fmtName[] = {"AU", "AIFF", "WAVE"};
    printf("outputFileFormat:\t%s\n", (p.outputFileFormat >= 0 && p.outputFileFormat <= 2) ? fmtName[p.outputFileFormat] : "Unknown");
    printf("outputRate:\t\t%.1f Hz\n", p.outputRate);
    printf("controlRate:\t\t%.2f Hz\n\n", p.controlRate);
    printf("volume:\t\t\t%.2f dB\n", p.volume);
    printf("channels:\t\t%-lu\n", (unsigned long)p.channels);
    printf("balance:\t\t%+1.2f\n\n", p.balance);
    printf("waveform:\t\t%s\n", p.waveform == 0 ? "Pulse" : p.waveform == 1 ? "Sine" : "Unknown");
    printf("tp:\t\t\t%.2f%%\n", p.tp);
    printf("tnMin:\t\t\t%.2f%%\n", p.tnMin);
    printf("tnMax:\t\t\t%.2f%%\n", p.tnMax);
    printf("breathiness:\t\t%.2f%%\n\n", p.breathiness);
    printf("nominal tube length:\t%.2f cm\n", p.length);
    printf("temperature:\t\t%.2f degrees C\n", p.temperature);
    printf("lossFactor:\t\t%.2f%%\n\n", p.lossFactor);
    printf("apScale:\t\t%.2f cm\n", p.apScale);
    printf("mouthCoef:\t\t%.1f Hz\n", p.mouthCoef);
    printf("noseCoef:\t\t%.1f Hz\n\n", p.noseCoef);
    for (long i = 1; i < TRM_TOTAL_NASAL_SECTIONS; i++) printf("n%-ld:\t\t\t%.2f cm\n", i, p.noseRadius[i]);
    printf("\nthroatCutoff:\t\t%.1f Hz\n", p.throatCutoff);
    printf("throatVol:\t\t%.2f dB\n\n", p.throatVol);
    printf("modulation:\t\t");
    printf("%s\n", p.usesModulation ? "on" : "off");
    printf("mixOffset:\t\t%.2f dB\n\n", p.mixOffset);
    // derived values, TRMTubeModel.m:599-602
    const trm_derived &d = t->b->d;
    printf("\nactual tube length:\t%.4f cm\n", d.actualTubeLength);
    printf("internal sample rate:\t%-d Hz\n", d.sampleRate);
    printf("control period:\t\t%-d samples (%.4f seconds)\n\n", d.controlPeriod, (float)d.controlPeriod / (float)d.sampleRate);
    // -[TRMDataList printControlRateInputTable], TRMDataList.m:294-330
    printf("\n%-lu control rate input tables:\n\n", (unsigned long)nframes);
    printf("glPitch\tglotVol\taspVol\tfricVol\tfricPos\tfricCF\tfricBW");
    for (unsigned long i = 0; i < TRM_TOTAL_REGIONS; i++) printf("\tr%-lu", i + 1);
    printf("\tvelum\n");
    for (size_t f = 0; f < nframes; f++) {
        const trm_parameters &q = frames[f];
        printf("%.2f\t%.2f\t%.2f\t%.2f\t%.2f\t%.2f\t%.2f", q.glottalPitch, q.glottalVolume, q.aspirationVolume, q.fricationVolume,
               q.fricationPosition, q.fricationCenterFrequency, q.fricationBandwidth);
        for (int i = 0; i < TRM_TOTAL_REGIONS; i++) printf("\t%.2f", q.radius[i]);
        printf("\t%.2f\n", q.velum);
    }
    printf("\n");
    fflush(stdout);
    return TRM_OK;
}

int trm_tube_save_output_to_file(trm_tube *t, const char *filename)
{
    if (!t || !filename) return fail(TRM_EINVAL, "null argument");
    // The reference prints these unconditionally (TRMTubeModel.m:372-376).
    double scale = (32767.0 / (double)t->maxSample) * trm::io_amplitude(t->b->params.volume);
    printf("\nnumber of samples:\t%-d\n", (int)t->numberSamples);
    printf("maximum sample value:\t%.4f\n", (double)t->maxSample);
    printf("scale:\t\t\t%.4f\n", scale);
    int rc = trm::io_write_sound_file(filename, t->b->params, t->samples.data(), t->numberSamples, (double)t->maxSample);
    if (rc) return fail(rc, "cannot write %s", filename);
    return TRM_OK;
}

int trm_write_sound_file(const trm_input_params *params, const float *samples, size_t n, float maximumSampleValue, const char *filename)
{
    if (!params || !filename || (n && !samples)) return fail(TRM_EINVAL, "null argument");
    int rc = trm::io_write_sound_file(filename, *params, samples, n, (double)maximumSampleValue);
    if (rc) return fail(rc, "cannot write %s", filename);
    return TRM_OK;
}

int trm_tube_generate_wav_data(trm_tube *t, uint8_t *buf, size_t cap, size_t *len)
{
    if (!t || !len) return fail(TRM_EINVAL, "null argument");
    if (t->maxSample == 0.f) return fail(TRM_ESILENT, "maximumSampleValue == 0 (TRMTubeModel.m:511)");
    size_t need = trm::io_wav_data_size(t->b->params, t->numberSamples);
    *len = need;
    if (!buf) return TRM_OK;
    if (cap < need) return fail(TRM_EINVAL, "buffer too small: %zu < %zu", cap, need);
    trm::io_wav_data(t->b->params, t->samples.data(), t->numberSamples, (double)t->maxSample, buf);
    return TRM_OK;
}

// ------------------------------------------------------------------ TRMDataList
int trm_data_list_read_file(const char *path, trm_input_params *params, trm_parameters **frames, size_t *nframes)
{
    if (!path || !params || !frames || !nframes) return fail(TRM_EINVAL, "null argument");
    std::vector<trm_parameters> v;
    int rc = trm::io_read_data_list(path, *params, v);
    if (rc) return fail(rc, "%s: %s", path, trm_strerror(rc));
    *nframes = v.size();
    *frames = nullptr;
    if (!v.empty()) {
        *frames = (trm_parameters *)malloc(v.size() * sizeof(trm_parameters));
        if (!*frames) return fail(TRM_ENOMEM, "frames");
        memcpy(*frames, v.data(), v.size() * sizeof(trm_parameters));
    }
    return TRM_OK;
}

int trm_data_list_write_file(const char *path, const trm_input_params *params, const trm_parameters *frames, size_t nframes)
{
    if (!path || !params || (nframes && !frames)) return fail(TRM_EINVAL, "null argument");
    int rc = trm::io_write_data_list(path, *params, frames, nframes);
    if (rc) return fail(rc, "%s: %s", path, trm_strerror(rc));
    return TRM_OK;
}

}  // extern "C"
