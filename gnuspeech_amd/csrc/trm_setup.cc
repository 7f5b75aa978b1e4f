// trm_setup.cc -- see trm_setup.h.  Host-side, once per tube/batch; double precision.
#include "trm_setup.h"

#include <math.h>
#include <string.h>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

namespace trm {

namespace {

constexpr int kSrcLen = 13 * 256;      // FILTER_LENGTH, TRMSampleRateConverter.m:20

double amplitude(double db)            // TRMUtility.m:26-41
{
    db -= 60.0;
    if (db <= -60.0) return 0.0;
    if (db >= 0.0) return 1.0;
    return pow(10.0, db / 20.0);
}

double izero(double x)                 // TRMUtility.m:50-66
{
    double sum = 1, u = 1, n = 1, halfx = x / 2.0;
    do {
        double t = halfx / n;
        n += 1;
        t *= t;
        u *= t;
        sum += u;
    } while (u >= 1E-21 * sum);
    return sum;
}

}  // namespace

// The oscillator's 49-tap FIR: TRMFIRFilter.h:7-9 fixes beta = 0.2, gamma = 0.1, cutoff = 1e-8, so the maximally-flat
// design (TRMFIRFilter.m:161-233, trimmed and mirrored :73-83, :236-244) has ONE outcome: c[0..24] below, c[48 - i] = c[i].
// The design routine itself lives in the oracle (oracle/trm_oracle.c: maximally_flat); tests/test_quad_model.py checks
// this table against the taps the reference binary computed (tests/golden/*.npz: firCoef).
const double kFirHalf[kFirUnique] = {
    1.0887157865533967e-08, 7.3133299513665035e-08, 2.137841904426972e-07, -6.5322947036563118e-08, -2.4300136304004587e-06,
    -6.267564867199802e-06, 2.2354964101677552e-06, 4.3261082316484397e-05, 6.9699584503931649e-05, -8.6968794272942906e-05,
    -0.00043243790948125553, -0.00028644112282104019, 0.0011316111860893143, 0.0023296835220954727, -0.00061607377823225985,
    -0.0070911561060204289, -0.0059227989476883834, 0.011204567485223504, 0.024723951583638423, -0.0013632376466076977,
    -0.056044214360539933, -0.051247637455809028, 0.087853092781387032, 0.2965041881371811, 0.39847427941239039};

void build_src_h(std::vector<double> &h, std::vector<double> &dh)
{
    h.assign(kSrcLen, 0.0);
    dh.assign(kSrcLen, 0.0);
    const double lp = 11.0 / 13.0, kaiser = 5.658;
    h[0] = lp;
    double x = M_PI / 256.0;
    for (int i = 1; i < kSrcLen; i++) {
        double y = (double)i * x;
        h[i] = sin(y * lp) / y;
    }
    double ib = 1.0 / izero(kaiser);
    for (int i = 0; i < kSrcLen; i++) {
        double t = (double)i / kSrcLen;
        h[i] *= izero(kaiser * sqrt(1.0 - t * t)) * ib;
    }
    for (int i = 0; i < kSrcLen - 1; i++) dh[i] = h[i + 1] - h[i];
    dh[kSrcLen - 1] = 0.0 - h[kSrcLen - 1];
}

void build_src_rows(std::vector<float> &rows)
{
    std::vector<double> h, dh;
    build_src_h(h, dh);
    rows.assign((size_t)65536 * kSrcRowC, 0.0f);
    // coefficient of tap i of a wing at 16-bit phase g (TRMSampleRateConverter.m:182-190)
    auto wing = [&](uint32_t g, int i) {
        uint32_t l = g >> 8, m = g & 255;
        uint32_t fi = l + 256u * (uint32_t)i;
        return (float)(h[fi] + dh[fi] * ((double)m / 256.0));
    };
    for (uint32_t f = 0; f < 65536; f++) {
        float *c = &rows[(size_t)f * kSrcRowC];
        for (int i = 0; i < kSrcWing; i++) {
            c[12 - i] = wing(f, i);                 // left wing, phase f, walks back from s[e-13]
            c[13 + i] = wing(0xFFFFu - f, i);       // right wing, phase ~f (:193-203), walks on from s[e-12]
        }
    }
}

void build_src_fine(std::vector<float> &fine)
{
    std::vector<double> h, dh;
    build_src_h(h, dh);
    fine.resize((size_t)kSrcLen * 256);
    for (uint32_t q = 0; q < (uint32_t)kSrcLen * 256u; q++)
        fine[q] = (float)(h[q >> 8] + dh[q >> 8] * ((double)(q & 255) / 256.0));
}

void build_down_rows(const std::vector<float> &fine, double ratio, uint32_t phaseIncrement, uint32_t &lmax,
                     uint32_t &rmax, uint32_t &pitch, std::vector<float> &rows)
{
    const uint32_t span = (uint32_t)kSrcLen * 256u;               // the wings end where (ph >> 8) reaches kSrcLen
    // both wings start at rint(f' * ratio) >= 0 for some 16-bit f': at most this many taps
    lmax = rmax = (span + phaseIncrement - 1) / phaseIncrement;
    pitch = (lmax + rmax + 3u) & ~3u;
    rows.assign((size_t)65536 * pitch, 0.0f);
    for (uint32_t f = 0; f < 65536u; f++) {
        float *row = &rows[(size_t)f * pitch];
        uint32_t ph = (uint32_t)rint((double)f * ratio);                              // :243
        for (uint32_t j = 0; j < lmax && (ph >> 8) < (uint32_t)kSrcLen; j++, ph += phaseIncrement) row[j] = fine[ph];
        ph = (uint32_t)rint((double)((~f) & 0xFFFFu) * ratio);                        // :257
        for (uint32_t j = 0; j < rmax && (ph >> 8) < (uint32_t)kSrcLen; j++, ph += phaseIncrement) row[lmax + j] = fine[ph];
    }
}

void build_sine_table(std::vector<float> &tab)
{
    tab.resize(kTableLen);
    for (int i = 0; i < kTableLen; i++) tab[i] = (float)sin(((double)i / (double)kTableLen) * 2.0 * M_PI);
}

uint64_t count_outputs(const trm_derived &d, uint64_t ntube)
{
    return src_count_outputs(ntube, (uint32_t)d.padSize, d.timeRegisterIncrement);      // trm_lane.h: the kernels use the same function
}

int build_const(const trm_input_params &p, Const &c, trm_derived &d)
{
    memset(&c, 0, sizeof c);
    memset(&d, 0, sizeof d);
    if (!(p.length > 0.0)) return TRM_EINVAL_LENGTH;                      // TRMTubeModel.m:197,204-207
    double speed = 331.4 + 0.6 * p.temperature;                          // TRMUtility.m:20-23
    d.controlPeriod = (int32_t)rint((speed * 10 * 100.0) / (p.length * p.controlRate));   // :200
    d.sampleRate = (int32_t)(p.controlRate * d.controlPeriod);           // :201 (float arithmetic)
    if (d.controlPeriod < 1 || d.sampleRate < 1) return TRM_ERANGE;
    d.actualTubeLength = (speed * 10 * 100.0) / d.sampleRate;            // :202
    double nyquist = (double)d.sampleRate / 2.0;
    d.sampleRateRatio = (double)p.outputRate / (double)d.sampleRate;     // TRMSampleRateConverter.m:80
    if (!(d.sampleRateRatio > 0.0)) return TRM_ERANGE;
    d.timeRegisterIncrement = (uint32_t)(int)rint(65536.0 / d.sampleRateRatio);            // :83
    if (d.timeRegisterIncrement == 0) return TRM_ERANGE;
    double rounded = 65536.0 / (double)d.timeRegisterIncrement;          // :86
    if (d.sampleRateRatio >= 1.0) {
        d.phaseIncrement = 0;
        d.padSize = 13;
    } else {
        d.phaseIncrement = (uint32_t)rint(d.sampleRateRatio * 65536.0);  // :92
        d.padSize = (int32_t)((float)13 / rounded) + 1;                  // :96
    }
    d.firTaps = kFirTaps;                                                // TRMFIRFilter.h:7-9: always the same 49

    c.controlPeriod = d.controlPeriod;
    c.sampleRate = d.sampleRate;
    c.waveform = p.waveform == TRM_WAVEFORM_PULSE ? 0 : 1;
    c.usesModulation = p.usesModulation != 0;
    c.invControlPeriod = (float)(1.0 / d.controlPeriod);
    c.invControlPeriodD = 1.0 / d.controlPeriod;
    c.damping = (float)(1.0 - p.lossFactor / 100.0);                     // :216
    c.breath = (float)(p.breathiness / 100.0);                           // :210
    c.crossmixFactor = (float)(1.0 / amplitude(p.mixOffset));            // :213
    double nk6 = 0.0;
    for (int i = 1; i < 5; i++) {                                        // :695-699
        double a2 = p.noseRadius[i] * p.noseRadius[i], b2 = p.noseRadius[i + 1] * p.noseRadius[i + 1];
        c.nasalK[i - 1] = (float)((a2 - b2) / (a2 + b2));
        c.nasalTd[i - 1] = (float)((a2 + a2) / (a2 + b2) * (1.0 - p.lossFactor / 100.0));   // (1 + k) d without cancellation
    }
    {
        double a2 = p.noseRadius[5] * p.noseRadius[5], b2 = p.apScale * p.apScale;   // :703-705
        c.nasalK[4] = (float)((a2 - b2) / (a2 + b2));
        c.onePlusNK6 = (float)(1.0 + (a2 - b2) / (a2 + b2));
        nk6 = (a2 - b2) / (a2 + b2);
    }
    c.noseR1sq = (float)(p.noseRadius[1] * p.noseRadius[1]);
    c.apScaleSq = (float)(p.apScale * p.apScale);
    {
        double coeff = (nyquist - p.mouthCoef) / nyquist;                // :222, TRMFilters.m:34-45
        c.mCoeff = (float)coeff; c.mA10 = (float)(1.0 - fabs(-coeff));
        coeff = (nyquist - p.noseCoef) / nyquist;                        // :225
        c.nCoeff = (float)coeff; c.nA10 = (float)(1.0 - fabs(-coeff));
        c.nasalK6a = (float)(nk6 * (1.0 - fabs(-coeff)));                // NC6 a10: the nose end's reflection gain (:848)
    }
    {
        double ta0 = (p.throatCutoff * 2.0) / d.sampleRate;              // :238
        c.ta0 = (float)ta0; c.tb1 = (float)(1.0 - ta0);
        c.throatGain = (float)amplitude(p.throatVol);                    // :239
    }
    c.invSampleRate = (float)(1.0 / d.sampleRate);
    c.fricGain = 1.0f;
    c.tableDiv1 = (int32_t)rint(512 * (p.tp / 100.0));                   // TRMWavetable.m:71-75
    c.tableDiv2 = (int32_t)rint(512 * ((p.tp + p.tnMax) / 100.0));
    c.invDiv1 = c.tableDiv1 > 0 ? (float)(1.0 / c.tableDiv1) : 0.0f;
    c.riseBias = c.tableDiv1 > 0 ? 0.0f : 1.0f;                          // no rise entries: the table is the fall alone (:81-96)
    c.tnDelta = rint(512 * ((p.tnMax - p.tnMin) / 100.0));
    c.basicIncrement = 512.0 / (double)d.sampleRate;
    for (int i = 0; i < kFirUnique; i++) c.fir[i] = (float)kFirHalf[i];
    c.timeRegisterIncrement = d.timeRegisterIncrement;
    c.phaseIncrement = d.phaseIncrement;
    c.padSize = d.padSize;
    c.upsample = d.sampleRateRatio >= 1.0;
    c.sampleRateRatioD = d.sampleRateRatio;
    if (c.tableDiv1 < 0 || c.tableDiv2 > 512 || c.tableDiv1 > c.tableDiv2) return TRM_ERANGE;
    // (tnMax < tnMin moves the closure point PAST tableDiv2 as the amplitude grows; beyond entry 512 the reference writes outside
    // its table, TRMWavetable.m:117-156)
    if (c.tnDelta < 0.0 && (double)c.tableDiv2 - c.tnDelta > 512.0) return TRM_ERANGE;
    return TRM_OK;
}

}  // namespace trm
