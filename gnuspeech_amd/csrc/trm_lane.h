// trm_lane.h -- the per-lane Tube Resonance Model: ONE tube voice per GPU lane.
//
// This is the arithmetic of -[TRMTubeModel synthesize]'s sample loop
// (Frameworks/Tube/TRMTubeModel.m:292-354) re-designed for 64-wide CDNA4 wavefronts.  A voice
// never leaves its lane, but the loop is cut into four stages that different waves of one
// workgroup run as a pipeline (the stage boundaries are exactly the feed-forward cuts of the
// reference's data flow):
//
//   excite   control tracks (pitch, voicing, aspiration) -> glottal oscillator -> 49-tap FIR ->
//            noise mixing  => {tract input, noise signal for the frication filter, throat input}
//   coef     control tracks (radii, velum, frication)  -> scattering coefficients, frication
//            taps, band-pass coefficients
//   tube     the recurrences: frication band-pass, 10+6 section waveguide, mouth/nose
//            reflection+radiation filters, throat low-pass  => one tube-rate sample
//   convert  band-limited sample-rate conversion to the output rate
//
// Everything that is identical for all voices of a batch (sample index, control-period position,
// converter phase, noise sequence, filter taps) is wave-uniform.  fp32 carries the signal; fp64 is
// used only where the reference has a discontinuity (oscillator phase wrap, rint() of the glottal
// closure point, (int) of the frication position, dB clamps) -- SURVEY.md 9.4.
//
// The same header compiles for the device (HIP, gfx950) and, for numerics tests only, for the
// host (tests/_emul).  No product path runs the host build.
#pragma once

#include <math.h>
#include <stdint.h>

#if defined(__HIP__)
#define TRM_HD __host__ __device__ inline __attribute__((always_inline))
#else
#define TRM_HD inline
#endif

namespace trm {

#ifndef TRM_WG_T
#define TRM_WG_T float
#endif
typedef TRM_WG_T wg_t;   // waveguide state type (fp32 in the product; tests may widen it to study rounding)

constexpr int kFirUnique = 25;      // 49 symmetric taps (TRMFIRFilter.h:7-9 design)
constexpr int kFirTaps = 49;
constexpr int kSrcWing = 13;        // ZERO_CROSSINGS (TRMSampleRateConverter.m:10)
constexpr int kSrcWindow = 26;
constexpr int kTableLen = 512;      // TRMWavetable.m:22
constexpr float kVtScale = 0.125f;  // TRMTubeModel.m:72

// ---------------------------------------------------------------- math primitives
#if defined(__HIP_DEVICE_COMPILE__)
TRM_HD float rcp_f(float x) { return __builtin_amdgcn_rcpf(x); }       // v_rcp_f32, 1 ulp
TRM_HD float exp2_f(float x) { return __builtin_amdgcn_exp2f(x); }     // v_exp_f32, args here are in [-20, 0]
TRM_HD float fma_f(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
TRM_HD double exp2_d(double x) { return exp2(x); }
TRM_HD double rint_d(double x) { return __builtin_rint(x); }
TRM_HD float rint_f(float x) { return __builtin_rintf(x); }
// a * b + c into a register of its OWN (v_fma_f32, three addresses), where the compiler would pick the two-address
// v_fmac_f32 and accumulate in c's register: for a shift register held in VGPRs (the transposed FIR's partial sums) that
// choice costs one v_mov per element and sample to move everything back at the loop's back edge
TRM_HD float fma_new_f(float a, float b, float c)
{
    float r;
    asm("v_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
// clamp to [0, 1]; NaN -> 0 (v_med3_f32 / the producing instruction's clamp modifier: a plain VGPR instruction where a
// compare + select pair costs a shared SIMD twice as much, profiles/valu_ceiling_r02.txt)
TRM_HD float sat_f(float x) { return __builtin_amdgcn_fmed3f(x, 0.0f, 1.0f); }
#else
TRM_HD float rcp_f(float x) { return 1.0f / x; }
TRM_HD float exp2_f(float x) { return exp2f(x); }
TRM_HD float fma_f(float a, float b, float c) { return fmaf(a, b, c); }
TRM_HD double fma_f(double a, double b, double c) { return fma(a, b, c); }   // (host studies with TRM_WG_T=double)
TRM_HD double exp2_d(double x) { return exp2(x); }
TRM_HD double rint_d(double x) { return rint(x); }
TRM_HD float rint_f(float x) { return rintf(x); }
TRM_HD float sat_f(float x) { return x > 0.0f ? (x < 1.0f ? x : 1.0f) : 0.0f; }      // (NaN -> 0 like the device's)
TRM_HD float fma_new_f(float a, float b, float c) { return fmaf(a, b, c); }
#endif

// sin / cos on [0, pi/4] (Taylor; truncation < 3e-8, below fp32 epsilon)
TRM_HD float sin_q(float y)
{
    float y2 = y * y;
    float p = fma_f(y2, 2.7557319e-6f, -1.9841270e-4f);
    p = fma_f(y2, p, 8.3333333e-3f);
    p = fma_f(y2, p, -1.6666667e-1f);
    return fma_f(y * y2, p, y);
}
TRM_HD float cos_q(float y)
{
    float y2 = y * y;
    float p = fma_f(y2, 2.4801587e-5f, -1.3888889e-3f);
    p = fma_f(y2, p, 4.1666667e-2f);
    p = fma_f(y2, p, -0.5f);
    return fma_f(y2, p, 1.0f);
}

// sin on [0, pi/2] (Taylor to x^13; truncation < 7e-10)
TRM_HD float sin_h(float y)
{
    float y2 = y * y;
    float p = fma_f(y2, 1.6059044e-10f, -2.5052108e-8f);
    p = fma_f(y2, p, 2.7557319e-6f);
    p = fma_f(y2, p, -1.9841270e-4f);
    p = fma_f(y2, p, 8.3333333e-3f);
    p = fma_f(y2, p, -1.6666667e-1f);
    return fma_f(y * y2, p, y);
}

// ---------------------------------------------------------------- wave-uniform constants
// Derived once per batch on the host from trm_input_params (TRMTubeModel.m:196-241).
struct Const {
    int32_t controlPeriod;
    int32_t sampleRate;
    int32_t waveform;           // 0 pulse, 1 sine
    int32_t usesModulation;
    float invControlPeriod;
    float damping;              // 1 - loss/100                              (:216)
    float breath;               // breathiness/100                           (:210)
    float crossmixFactor;       // 1/amplitude(mixOffset)                    (:213)
    float nasalK[5];            // NC2..NC6, fixed                           (:692-707)
    float nasalTd[4];           // (1 + NC2..NC5) times damping, formed in double without cancellation: the junctions'
                                // working form (tube_step)
    float onePlusNK6;           // 1 + NC6, formed in double                 (:849)
    float nasalK6a;             // NC6 times the nose reflection filter's a10 (:848, TRMFilters.m:47-52)
    float noseR1sq;             // noseRadius[1]^2, for NC1                  (:741)
    float apScaleSq;            // apScale^2, for C8                         (:724)
    // mouth / nose reflection+radiation pairs (TRMFilters.m:34-45): a20 = coeff, a21 = b21 = b11 =
    // -coeff, a10 = 1 - |coeff|
    float mCoeff, nCoeff;       // (mouth, nose) pairs: the four-lane kernel filters both ends as one two-wide value
    float mA10, nA10;
    float ta0, tb1, throatGain;           // throat low-pass                 (TRMFilters.m:64-68)
    float invSampleRate;
    float fricGain;             // 1 (:750); 10 in TRAcT's loop order (Applications/TRAcT/tube.c:1371, trm_stream_set_mode)
    // glottal pulse table geometry (TRMWavetable.m:71-75)
    int32_t tableDiv1, tableDiv2;
    float invDiv1;
    float riseBias;             // 0; 1 when tableDiv1 == 0: the rise has no entries (TRMWavetable.m:81 loops zero times) and
                                // its factor in pulse_table_f must be exactly 1 everywhere
    double tnDelta;
    double basicIncrement;      // 512 / sampleRate
    double invControlPeriodD;
    float fir[kFirUnique];      // c[0..24]; c[48-i] == c[i]
    // sample-rate converter (TRMSampleRateConverter.m:80-98)
    uint32_t timeRegisterIncrement;
    uint32_t phaseIncrement;    // down-sampling only
    int32_t padSize;
    int32_t upsample;           // sampleRateRatio >= 1
    double sampleRateRatioD;
};

// ================================================================ stage 1: excitation
// Control-period interpolation state of the columns this stage consumes (TRMTubeModel.m:611-688).
struct ExciteTrack {
    double f0, f0Ratio;         // 220*2^((pitch+3)/12) as a geometric sequence per sample
    double glotDb, glotDbDelta; // dB value, repeated addition like the reference
    double axGeo, axRatio;      // 10^((dB-60)/20) as a geometric sequence
    float aspBase, aspDelta;    // aspiration volume, dB
};

struct ExciteState {
    double oscPos;              // wavetable position (TRMWavetable.m:165-168)
    float fir[24];              // transposed-form partial sums of the 49-tap FIR
};

struct Excitation {
    float gin;                  // (pulse + ah1*signal) * VT_SCALE : tract input      (:336)
    float sig;                  // noise signal fed to the frication band-pass        (:337)
    float thr;                  // pulse * VT_SCALE : throat input                    (:341)
};

TRM_HD void excite_reset(ExciteState &S)
{
    S.oscPos = 0.0;
    for (int i = 0; i < 24; i++) S.fir[i] = 0.f;
}

// -setControlRateParameters:previous: (TRMTubeModel.m:611-672), columns pitch / glotVol / aspVol.
TRM_HD void excite_track_setup(ExciteTrack &T, const Const &C, const float *prev, const float *cur)
{
    const double kLog2_10_over_20 = 0.16609640474436813;   // log2(10)/20
    double p0 = (double)prev[0], dp = ((double)cur[0] - p0) * C.invControlPeriodD;
    T.f0 = 220.0 * exp2_d((p0 + 3.0) * (1.0 / 12.0));      // frequency(), TRMUtility.m:44-47
    T.f0Ratio = exp2_d(dp * (1.0 / 12.0));
    double v0 = (double)prev[1], dv = ((double)cur[1] - v0) / (double)C.controlPeriod;
    T.glotDb = v0;
    T.glotDbDelta = dv;
    T.axGeo = exp2_d((v0 - 60.0) * kLog2_10_over_20);      // amplitude(), TRMUtility.m:26-41
    T.axRatio = exp2_d(dv * kLog2_10_over_20);
    T.aspBase = prev[2];
    T.aspDelta = (cur[2] - prev[2]) * C.invControlPeriod;
}

// Entry i of the 512-entry sine table (TRMWavetable.m:98-101), computed instead of stored: sin(2*pi*i/512)
// by quadrant folding onto the [0, pi/4] polynomials (error < 1e-7, the table itself is fp32 here).
TRM_HD float sine_table(int i)
{
    int q = i & 127;                       // position inside the quadrant
    int quad = (i >> 7) & 3;
    int m = (quad & 1) ? 128 - q : q;      // odd quadrants run backwards
    bool useCos = m > 64;                  // fold [pi/4, pi/2] onto cos of the complement
    float y = (float)(useCos ? 128 - m : m) * (6.28318530717959f / 512.0f);
    float v = useCos ? cos_q(y) : sin_q(y);
    return quad >= 2 ? -v : v;
}

// dB -> linear amplitude with the reference's clamps (TRMUtility.m:26-41), fp32.
TRM_HD float amplitude_f(float db)
{
    // the clamps as arithmetic: 2^x >= 1 exactly from db = 60 on, so "db >= 60 -> 1" is min(2^x, 1); "db <= 0 -> 0" is a
    // factor that is 0 there and exactly 1 from db = 2^-100 on
    const float a = sat_f(exp2_f((db - 60.0f) * 0.16609640474f));
    return a * sat_f(db * 1.2676506e30f);
}

// One entry of the glottal pulse table as a pure function of the closure point
// (TRMWavetable.m:79-96 rise/closed, :117-156 fall rewritten from the amplitude).
// The three regions without a compare or a select: with x and xf clamped to [0, 1] the rise polynomial is exactly 1 from
// the end of the rise on and the fall exactly 1 before its start, so the entry is their PRODUCT (one factor is always
// exactly 1: the product is the other one, bit for bit); the closed phase is the fall's clamp at 1.  fi = the entry's index as a float.
TRM_HD float pulse_table_f(float fi, float riseBias, float invDiv1, float fNewDiv2, float invFall)
{
    // (riseBias is 0 -- the FMA is then the exact product fi / div1 -- except for a pulse without a rise, tp ~ 0, where it is 1)
    const float x = sat_f(fma_f(fi, invDiv1, riseBias));
    const float rise = x * x * fma_f(-2.0f, x, 3.0f);
    // the fall's abscissa measured from its END: exactly 1 from newDiv2 on (the closed phase: fall = 0 exactly, no separate
    // factor), below 0 -> 0 before the fall starts; at its first entry it is 0 or one ulp, which 1 - xf^2 does not see.
    // invFall = 1 / max(newDiv2 - div1, 1/2): a fall of no length closes at once (osc_read)
    const float xf = sat_f(fma_f(fi - fNewDiv2, invFall, 1.0f));
    const float fall = fma_f(-xf, xf, 1.0f);
    return rise * fall;
}
// The excitation stage in two halves (they may run in different waves):
//   osc_sample  control tracks + 2x oversampled wavetable oscillator -> the two table reads of the
//               sample and the two amplitudes                              (fp64 tracks live here)
//   mix_sample  49-tap FIR + noise mixing -> Excitation                    (FIR state lives here)
struct OscOut {
    float wa, wb;               // the two oversampled wavetable reads (TRMWavetable.m:178-190)
    float ax, ah1;              // amplitude(glotVol), amplitude(aspVol) (TRMTubeModel.m:295-296)
};

struct OscState {
    double oscPos;              // wavetable position (TRMWavetable.m:165-168)
};

struct FirState {
    float fir[24];              // transposed-form partial sums of the 49-tap FIR
};

// The two oversampled wavetable reads at positions pos1, pos2 for the amplitude of voicing axd
// (TRMWavetable.m:117-195): the table is evaluated, not stored.
template <class SineLookup>
TRM_HD void osc_read(const Const &C, double axd, double pos1, double pos2, SineLookup sineTab, float &wa, float &wb)
{
    int lo1 = (int)pos1, lo2 = (int)pos2;           // 0 .. 511: the position lies in (-1, 511] and the cast truncates (:183)
    float fr1 = (float)(pos1 - (double)lo1), fr2 = (float)(pos2 - (double)lo2);
    float a0, a1, b0, b1;
    if (C.waveform == 0) {
        // newDiv2 = tableDiv2 - rint(amplitude * tnDelta) (:122), an integer kept in a float
        const float fDiv1 = (float)C.tableDiv1, fNew = (float)C.tableDiv2 - (float)rint_d(axd * C.tnDelta);
        const float invFall = rcp_f(fmaxf(fNew - fDiv1, 0.5f));
        // the upper entry is entry lo + 1 without the wrap: "entry 512" evaluates to entry 0's value, 0 (rise 1, fall closed:
        // newDiv2 <= tableDiv2 <= 512, build_const)
        const float f1 = (float)lo1, f2 = (float)lo2;
        a0 = pulse_table_f(f1, C.riseBias, C.invDiv1, fNew, invFall);
        a1 = pulse_table_f(f1 + 1.0f, C.riseBias, C.invDiv1, fNew, invFall);
        b0 = pulse_table_f(f2, C.riseBias, C.invDiv1, fNew, invFall);
        b1 = pulse_table_f(f2 + 1.0f, C.riseBias, C.invDiv1, fNew, invFall);
    } else {
        const int up1 = (lo1 + 1) & (kTableLen - 1), up2 = (lo2 + 1) & (kTableLen - 1);          // mod0(lower + 1), :185
        a0 = sineTab(lo1); a1 = sineTab(up1); b0 = sineTab(lo2); b1 = sineTab(up2);
    }
    wa = fma_f(fr1, a1 - a0, a0);
    wb = fma_f(fr2, b1 - b0, b0);
}

// `j` = position in the control period (uniform), `sineTab` = 512-entry sine table lookup.
// The oscillator's advance per half tube sample, f0/2 * 512/sampleRate table entries (TRMWavetable.m:178-181), rounded
// to a multiple of 2^-30 entries.  Every phase the kernels form is a sum of such increments below 2^11, hence EXACT in
// fp64: the four-lane kernel's prefix sums, the one-lane kernel's running sum and a stream cut anywhere arrive at the
// same positions bit for bit, whatever the order of the additions.  (2^-31 entries of rounding per sample: a random
// walk of 1e-6 entries over a million samples, far inside the tolerance.)
TRM_HD double osc_increment(double f0, const Const &C)
{
    return rint_d(((f0 * 0.5) * C.basicIncrement) * 1073741824.0) * (1.0 / 1073741824.0);
}

template <class SineLookup>
TRM_HD OscOut osc_sample(OscState &S, ExciteTrack &T, const Const &C, int j, SineLookup sineTab)
{
    // amplitude of voicing with its clamps (:294-296), in fp64: it feeds rint() below
    double axd = T.glotDb >= 60.0 ? 1.0 : T.axGeo;
    axd = T.glotDb <= 0.0 ? 0.0 : axd;
    OscOut O;
    O.ax = (float)axd;
    O.ah1 = amplitude_f(fma_f((float)j, T.aspDelta, T.aspBase));

    // glottal source: 2x oversampled wavetable oscillator (TRMWavetable.m:117-195)
    double inc = osc_increment(T.f0, C);
    double pos1 = S.oscPos + inc;
    pos1 = pos1 > 511.0 ? pos1 - 512.0 : pos1;          // mod0(), :28-34
    double pos2 = pos1 + inc;
    pos2 = pos2 > 511.0 ? pos2 - 512.0 : pos2;
    S.oscPos = pos2;
    osc_read(C, axd, pos1, pos2, sineTab, O.wa, O.wb);
    // advance the fp64 tracks (:351)
    T.glotDb += T.glotDbDelta;
    T.axGeo *= T.axRatio;
    T.f0 *= T.f0Ratio;
    return O;
}

// source mixing (TRMTubeModel.m:315-341) from the FIR output `pulse`
TRM_HD Excitation mix_tail(const Const &C, float ax, float ah1, float pulse, float lpNoise)
{
    float pulsedNoise = lpNoise * pulse;
    pulse = ax * fma_f(pulsedNoise, C.breath, pulse * (1.0f - C.breath));
    float sig;
    if (C.usesModulation) {
        float cm = ax * C.crossmixFactor;
        cm = cm < 1.0f ? cm : 1.0f;
        sig = fma_f(pulsedNoise, cm, lpNoise * (1.0f - cm));
    } else
        sig = lpNoise;
    Excitation E;
    E.gin = fma_f(ah1, sig, pulse) * kVtScale;
    E.sig = sig;
    E.thr = pulse * kVtScale;
    return E;
}

// `fir` = the 25 distinct taps (the caller decides where they live), `lpNoise` = the voice-independent
// low-passed noise sample (uniform).
TRM_HD Excitation mix_sample(FirState &S, const Const &C, const float *fir, const OscOut &O, float lpNoise)
{
    // 49-tap FIR, decimate by 2, transposed form: y[m] = sum c[2k] b[m-k] + c[2k+1] a[m-k]
    // (TRMFIRFilter.m:116-146); the partial sums shift for free through the FMA destination.
    auto c = [&](int i) { return fir[i < kFirUnique ? i : (kFirTaps - 1) - i]; };
    float pulse = fma_f(c(0), O.wb, fma_f(c(1), O.wa, S.fir[0]));
    // (ascending q: element q's register is free when its new value is formed from element q + 1 -- no copies)
    for (int q = 0; q < 23; q++) S.fir[q] = fma_f(c(2 * q + 2), O.wb, fma_new_f(c(2 * q + 3), O.wa, S.fir[q + 1]));
    S.fir[23] = c(48) * O.wb;
    return mix_tail(C, O.ax, O.ah1, pulse, lpNoise);
}

// Both halves in one call (host emulation).
template <class SineLookup>
TRM_HD Excitation excite_sample(ExciteState &S, ExciteTrack &T, const Const &C, const float *fir, int j, float lpNoise,
                                SineLookup sineTab)
{
    OscState os; os.oscPos = S.oscPos;
    FirState fs;
    for (int i = 0; i < 24; i++) fs.fir[i] = S.fir[i];
    OscOut O = osc_sample(os, T, C, j, sineTab);
    Excitation E = mix_sample(fs, C, fir, O, lpNoise);
    S.oscPos = os.oscPos;
    for (int i = 0; i < 24; i++) S.fir[i] = fs.fir[i];
    return E;
}

// ================================================================ stage 2: coefficients
struct CoefTrack {
    float fricPos0, fricPosDelta;
    // fp32 base + delta: fricVol, fricCF, fricBW, r1..r8, velum
    float base[12], delta[12];
};

struct Coefs {
    float k8a;                  // C8 times the mouth reflection filter's a10: what the end filter multiplies by (:723-725, :820)
    float td[7], ntd1;          // (1 + C1..C7), (1 + NC1) times damping, formed without cancellation: what the
                                // junctions multiply by (tube_step)         (:712-722, :738-743)
    float onePlusK8;            // 1 + C8 without cancellation              (:835)
    float alphaLR, alphaU;      // three-way junction                       (:730-736)
    float tap[8];               // frication taps FC1..FC8                  (:748-773)
    float bpAlpha, bpBeta, bpGamma;   // frication band-pass                (TRMFilters.m:9-17)
    float pad_;
};

// frame columns: 3 fricVol, 4 fricPos, 5 fricCF, 6 fricBW, 7..14 radii, 15 velum
TRM_HD void coef_track_setup(CoefTrack &T, const Const &C, const float *prev, const float *cur)
{
    T.fricPos0 = prev[4];
    T.fricPosDelta = (cur[4] - prev[4]) * C.invControlPeriod;
    const int col[12] = {3, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15};
    for (int i = 0; i < 12; i++) {
        T.base[i] = prev[col[i]];
        T.delta[i] = (cur[col[i]] - prev[col[i]]) * C.invControlPeriod;
    }
}

// Stateless in the sample index: any wave may compute any sample j of the current control period.
// Two halves that touch disjoint fields of Coefs (they may run in different waves):
//   coef_sample_area   radii, velum -> scattering coefficients, three-way junction, NC1
//   coef_sample_fric   frication volume / position / band -> taps, band-pass coefficients
// CT: where the stage's wave-uniform constants come from -- `Const` (kernel arguments: scalar registers) or a CoefConst of
// vector-register copies (cf. TubeConst): an instruction with a scalar operand issues at the 4.3-cycle rate on a shared SIMD,
// a plain one at 2.4 (tools/ubench/valu_ceiling.hip).
struct CoefConst {
    float damping, apScaleSq, mA10, noseR1sq, invSampleRate, fricGain;
};

template <class CT>
TRM_HD void coef_sample_area(Coefs &K, const CoefTrack &T, const CT &C, int j)
{
    const float fj = (float)j;
    // control-rate interpolation (:676-688 evaluated as base + j*delta)
    float r2[8];
    for (int i = 0; i < 8; i++) {
        float r = fma_f(fj, T.delta[3 + i], T.base[3 + i]);
        r2[i] = r * r;
    }
    float velum = fma_f(fj, T.delta[11], T.base[11]);

    // scattering coefficients (:712-744) as transmission factors (1 + k) d = 2 a^2 d / (a^2 + b^2): no
    // cancellation when a junction nearly closes (k -> -1), see tube_step
    const float d2 = C.damping + C.damping;
    for (int i = 0; i < 7; i++) K.td[i] = r2[i] * (d2 * rcp_f(r2[i] + r2[i + 1]));
    float rk8 = rcp_f(r2[7] + C.apScaleSq);
    K.k8a = ((r2[7] - C.apScaleSq) * rk8) * C.mA10;
    K.onePlusK8 = (r2[7] + r2[7]) * rk8;         // 1 + C8 without the cancellation of a nearly closed mouth
    float v2 = velum * velum;
    float jsum = 2.0f * rcp_f(r2[3] + r2[3] + v2);
    K.alphaLR = jsum * r2[3];
    K.alphaU = jsum * v2;
    K.ntd1 = v2 * (d2 * rcp_f(v2 + C.noseR1sq));
}

// kFricGain: multiply the frication amplitude by C.fricGain (the streaming kernel instance only: trm_stream_set_mode;
// the one-shot instances carry no such instruction)
// kSatTaps: "max(., 0)" as the [0, 1] clamp of the FMA that forms the tap -- two operations per tap instead of three; only
// without the gain (the amplitude is at most 1 then) and only where it pays: the one-voice-per-lane kernel (8 of its 450
// instructions per sample); the eight-lane kernel's frication wave ran 3 % SLOWER with it (profiles/ab_r03.txt).
template <bool kFricGain = false, bool kSatTaps = false, class CT = Const>
TRM_HD void coef_sample_fric(Coefs &K, const CoefTrack &T, const CT &C, int j)
{
    const float fj = (float)j;
    float fricDb = fma_f(fj, T.delta[0], T.base[0]);
    float fricCF = fma_f(fj, T.delta[1], T.base[1]);
    float fricBW = fma_f(fj, T.delta[2], T.base[2]);
    // frication taps (:748-773).  The reference splits the position into (int) and fraction and gives the two taps
    // around it (1 - frac) amp and frac amp: the values of ONE continuous function, tap[i] = amp max(0, 1 - |pos - i|)
    // (linear interpolation between the taps; position 7.x: the second tap would be FC9, which does not exist, :761).
    // Evaluated that way there is no discontinuity for fp32 to miss (the (int) cast disappears), no fp64 and no
    // select chain: three plain operations per tap.  Positions below 0 are outside the tract (the reference's split
    // is meaningless there as well).
    float fricAmp = amplitude_f(fricDb);
    if (kFricGain) fricAmp *= C.fricGain;
    const float fricPos = fma_f(fj, T.fricPosDelta, T.fricPos0);                 // (:676-688)
    for (int i = 0; i < 8; i++) {
        const float dist = fabsf(fricPos - (float)i);
        const float t = fma_f(-fricAmp, dist, fricAmp);
        K.tap[i] = kSatTaps ? sat_f(t) : fmaxf(t, 0.0f);
    }

    // band-pass coefficients (TRMFilters.m:9-17): beta = (1 - t) / (2 (1 + t)), t = tan(pi BW / SR); gamma = (1/2 + beta)
    // cos(2 pi CF / SR); alpha = (1/2 - beta) / 2.
    {
        // (1 - tan x) / (1 + tan x) = tan(pi/4 - x): ONE tangent, and for every bandwidth the filter is stable at
        // (0 <= BW <= SR/2) its argument lies in [-pi/4, pi/4], where the two polynomials hold -- no quadrant folding, no
        // selects.  (Outside that range the reference's own output grows without bound: tests/cases.py bandpass_unstable.)
        float v = fricBW * C.invSampleRate;
        v = v - rint_f(v);                      // tan has period 1 in v
        const float y = 3.14159265358979f * (0.25f - v);
        K.bpBeta = (0.5f * sin_q(y)) * rcp_f(cos_q(y));
        // cos(2 pi u) = 1 - 2 sin^2(pi u), and with u reduced to [-1/2, 1/2] the sine's argument lies in [0, pi/2]: one
        // odd polynomial (Taylor to x^13: truncation 7e-10), again without folding
        float u = fricCF * C.invSampleRate;
        u = u - rint_f(u);
        const float sh = sin_h(3.14159265358979f * fabsf(u));
        const float cv = fma_f(-2.0f * sh, sh, 1.0f);
        K.bpGamma = (0.5f + K.bpBeta) * cv;
        K.bpAlpha = (0.5f - K.bpBeta) * 0.5f;
    }
    K.pad_ = 0.0f;
}

template <bool kFricGain = false, class CT = Const>
TRM_HD Coefs coef_sample(const CoefTrack &T, const CT &C, int j)
{
    Coefs K;
    coef_sample_area(K, T, C, j);
    coef_sample_fric<kFricGain, !kFricGain, CT>(K, T, C, j);
    return K;
}

// ================================================================ stage 3: the tube recurrences
// Travelling-wave values of one sample (TRMTubeModel.m:161-165).  The reference double-buffers them
// ([section][top|bottom][ping|pong]); here a step reads one Waves and writes another, so two
// consecutive samples ping-pong between two register sets without copies.
struct Waves {
    wg_t oT[10], oB[10];        // oropharynx top / bottom
    wg_t nT[6], nB[6];          // nasal
};

struct TubeFilters {
    wg_t mReflY, mRadX, mRadY;  // mouth filter memories
    wg_t nReflY, nRadX, nRadY;  // nose filter memories
    wg_t throatY;
    float bpX1, bpX2, bpY1, bpY2;   // frication band-pass memory (TRMFilters.m:19-29)
};

struct TubeState {
    Waves w;
    TubeFilters f;
};

TRM_HD void waves_reset(Waves &W)
{
    for (int i = 0; i < 10; i++) { W.oT[i] = 0.f; W.oB[i] = 0.f; }
    for (int i = 0; i < 6; i++) { W.nT[i] = 0.f; W.nB[i] = 0.f; }
}

TRM_HD void filters_reset(TubeFilters &L)
{
    L.mReflY = L.mRadX = L.mRadY = 0.f;
    L.nReflY = L.nRadX = L.nRadY = 0.f;
    L.throatY = 0.f;
    L.bpX1 = L.bpX2 = L.bpY1 = L.bpY2 = 0.f;
}

TRM_HD void tube_reset(TubeState &S)
{
    waves_reset(S.w);
    filters_reset(S.f);
}

// throat low-pass (:341, TRMFilters.m:72-77): y = ta0 x + tb1 y1
TRM_HD float throat_filter(float y1, float ta0, float tb1, float x) { return fma_f(ta0, x, tb1 * y1); }

// One sample: old waves `o` -> new waves `nw` (all new values from old values only, :778-853).
// Returns the tube-rate output sample (what the reference hands to -dataFill:, :346).
// CT: where the step's wave-uniform constants come from -- `Const` itself (kernel arguments: scalar registers) or a
// TubeConst the caller filled with VECTOR-register copies (the one-voice-per-lane kernel: an instruction with a scalar
// operand costs a shared SIMD 4.3 cycles instead of 2.4, profiles/valu_ceiling_r02.txt, and the damping factor alone
// sits in half of the step's instructions).
struct TubeConst {
    float damping, mCoeff, nCoeff, nasalTd[4], nasalK6a, onePlusNK6, ta0, tb1, throatGain;
};

// kThroatDone: E.thr already IS the throat filter's output (the one-voice-per-lane kernel runs that one-pole section, which
// depends on the excitation alone, in its mixing wave: two instructions off the wave every step waits for; same operations,
// same bits).
template <class CT, bool kThroatDone = false>
TRM_HD float tube_step(const Waves &o, Waves &nw, TubeFilters &L, const CT &C, const Excitation &E, const Coefs &K)
{
    // frication band-pass (TRMFilters.m:19-29), evaluated before the tract (:336-337)
    float fric = 2.0f * fma_f(K.bpAlpha, E.sig - L.bpX2, fma_f(K.bpGamma, L.bpY1, -(K.bpBeta * L.bpY2)));
    L.bpX2 = L.bpX1; L.bpX1 = E.sig; L.bpY2 = L.bpY1; L.bpY1 = fric;

    // Scattering junction between a (top wave from the left) and b (bottom wave from the right), (:783-816):
    // T = (a + k (a - b)) d + inj, B = (b + k (a - b)) d.  With t = (1 + k) d formed by the coefficient stage WITHOUT
    // cancellation this is T = d b + t (a - b) + inj, B = T - d (a - b): the same four operations as the literal form
    // k d (a - b), but a nearly closed junction (k -> -1: a velum or a constriction that leaks into a quiet cavity)
    // transmits t a with the relative accuracy of t instead of rounding it off against d a -- the literal form in
    // fp32 put 2e-5 of error on a voice whose output is mostly such a leak (monet_vowel: closed mouth, velum 0.1).
    const float *td = K.td, *tap = K.tap;
    const wg_t d = C.damping;
    const wg_t input = E.gin;
    const wg_t fr = fric;
    auto junction = [&](float t, wg_t a, wg_t b, wg_t &T, wg_t &B) {
        const wg_t df = a - b;
        T = fma_f((wg_t)t, df, d * b);
        B = fma_f(-d, df, T);
    };
    nw.oT[0] = o.oB[0] * d + input;
    junction(td[0], o.oT[0], o.oB[1], nw.oT[1], nw.oB[0]);
    for (int i = 1; i < 3; i++) {             // S2-S3, S3-S4 with taps FC1, FC2
        junction(td[i], o.oT[i], o.oB[i + 1], nw.oT[i + 1], nw.oB[i]);
        nw.oT[i + 1] += tap[i - 1] * fr;
    }
    {
        // (written out as FMAs: left to the compiler, a sum of two products is fused one way or the other depending on
        // the code around it, and the kernels' two inlined copies of this step -- even / odd samples -- must round alike:
        // a stream cut at an odd sample count runs every later sample through the other copy)
        wg_t jp = fma_f((wg_t)K.alphaLR, o.oT[3], fma_f((wg_t)K.alphaLR, o.oB[4], K.alphaU * o.nB[0]));
        nw.oB[3] = (jp - o.oT[3]) * d;
        nw.oT[4] = (jp - o.oB[4]) * d + tap[2] * fr;
        nw.nT[0] = (jp - o.nB[0]) * d;
    }
    junction(td[3], o.oT[4], o.oB[5], nw.oT[5], nw.oB[4]);
    nw.oT[5] += tap[3] * fr;
    junction(C.damping, o.oT[5], o.oB[6], nw.oT[6], nw.oB[5]);      // the junction-less S6|S7 boundary: k = 0
    nw.oT[6] += tap[4] * fr;
    for (int i = 6; i < 9; i++) {             // S7-S8, S8-S9, S9-S10 with taps FC6..FC8
        junction(td[i - 2], o.oT[i], o.oB[i + 1], nw.oT[i + 1], nw.oB[i]);
        nw.oT[i + 1] += tap[i - 1] * fr;
    }
    wg_t out;
    {   // mouth: reflection y = a10*x - b11*y1, radiation y = a20*x + a21*x1 - b21*y1 (TRMFilters.m:47-60)
        wg_t refl = fma_f((wg_t)K.k8a, o.oT[9], C.mCoeff * L.mReflY);
        L.mReflY = refl;
        nw.oB[9] = d * refl;
        wg_t rin = K.onePlusK8 * o.oT[9];
        wg_t rad = C.mCoeff * (rin - L.mRadX + L.mRadY);
        L.mRadX = rin; L.mRadY = rad;
        out = rad;
    }
    {
        float tt[5] = {K.ntd1, C.nasalTd[0], C.nasalTd[1], C.nasalTd[2], C.nasalTd[3]};
        for (int i = 0; i < 5; i++) junction(tt[i], o.nT[i], o.nB[i + 1], nw.nT[i + 1], nw.nB[i]);
        wg_t refl = fma_f((wg_t)C.nasalK6a, o.nT[5], C.nCoeff * L.nReflY);
        L.nReflY = refl;
        nw.nB[5] = d * refl;
        wg_t rin = C.onePlusNK6 * o.nT[5];
        wg_t rad = C.nCoeff * (rin - L.nRadX + L.nRadY);
        L.nRadX = rin; L.nRadY = rad;
        out += rad;
    }
    // throat (:341, TRMFilters.m:72-77)
    wg_t ty = (wg_t)E.thr;
    if (!kThroatDone) {
        ty = throat_filter(L.throatY, C.ta0, C.tb1, E.thr);
        L.throatY = ty;
    }
    out = ty * C.throatGain + out;
    return (float)out;
}

// Single-sample form (host emulation): step into a scratch set, then commit.
TRM_HD float tube_sample(TubeState &S, const Const &C, const Excitation &E, const Coefs &K)
{
    Waves nw;
    float y = tube_step(S.w, nw, S.f, C, E, K);
    S.w = nw;
    return y;
}

// ================================================================ stage 4: sample-rate conversion
// One up-sampled output (TRMSampleRateConverter.m:171-233): 13 left + 13 right taps over the 26
// tube-rate samples w[0..25] = s[e-25 .. e], e = the converter's read position for this output.
// c[0..25] is the phase's combined coefficient row: c[i] = left-wing coefficient 12-i for i < 13,
// right-wing coefficient i-13 for i >= 13, so the output is one straight dot product.  Even and odd
// terms accumulate separately (two-wide packed FMAs on the device).
constexpr int kSrcRowC = 32;        // combined row: 26 coefficients + 6 pad = 128 bytes per phase

TRM_HD float src_dot(const float *w, const float *c)
{
    float a0 = w[0] * c[0], a1 = w[1] * c[1];
    for (int i = 2; i < kSrcWindow; i += 2) {
        a0 = fma_f(w[i], c[i], a0);
        a1 = fma_f(w[i + 1], c[i + 1], a1);
    }
    return a0 + a1;
}

// The same dot product over a 16-byte aligned window: w[0..31] starts `o` (0..3) samples before the
// output's first tap and c[0..31] is the coefficient row shifted right by `o` (zeros around it), so the
// extra terms contribute exactly 0.  Four partial sums, two-wide (packed FMAs on the device).
TRM_HD float src_dot32(const float *w, const float *c)
{
    float a0 = w[0] * c[0], a1 = w[1] * c[1], a2 = w[2] * c[2], a3 = w[3] * c[3];
    for (int i = 4; i < 32; i += 4) {
        a0 = fma_f(w[i], c[i], a0);
        a1 = fma_f(w[i + 1], c[i + 1], a1);
        a2 = fma_f(w[i + 2], c[i + 2], a2);
        a3 = fma_f(w[i + 3], c[i + 3], a3);
    }
    return (a0 + a2) + (a1 + a3);
}

// Converter bookkeeping in closed form (TRMSampleRateConverter.m:221-232): output k sits at input
// time k*inc (16.16 fixed point): phase = low 16 bits, read position e = high part.
TRM_HD uint32_t src_phase(uint32_t k, uint32_t inc) { return (k * inc) & 0xFFFFu; }
// Outputs of a voice of `ntube` tube samples: output k sits at input time k*inc and is produced while its read
// position floor(k*inc / 2^16) lies before the end of the data, ntube + 2*pad with the flush's zeros
// (TRMSampleRateConverter.m:160-173, TRMRingBuffer.m:85-93).  One reference behaviour belongs to the count: when
// DOWN-sampling (inc > 2^16) the read position advances by more than one sample per output, so a dataEmpty can stop
// one or two samples PAST its end pointer; if the flush's final dataEmpty then finds its own end pointer at or behind
// that position -- the data ended within a sample or two after a multiple of the ring's fill size, 1024 - 2*pad, where
// the previous dataEmpty ran -- it takes the end pointer for wrapped (:160-163) and converts one more lap of the ring:
// 1024 further input positions whose samples are whatever the ring still holds (src_ring_sample below).
constexpr uint32_t kSrcRing = 1024;        // TRMRingBuffer.h BUFFER_SIZE
TRM_HD uint64_t src_count_outputs(uint64_t ntube, uint32_t pad, uint32_t inc)
{
    const uint64_t total = ntube + 2ull * pad;
    if (inc > 65536u) {
        const uint64_t fill = kSrcRing - 2ull * pad;
        if (total >= fill) {
            const uint64_t prevEnd = (total / fill) * fill;                     // where the dataEmpty before the last one ended
            const uint64_t k = (prevEnd * 65536ull + inc - 1) / inc;            // the first output it left for the last one
            if (((k * inc) >> 16) > total) return ((total + kSrcRing) * 65536ull + inc - 1) / inc;
        }
    }
    return (total * 65536ull + inc - 1) / inc;
}
// What the converter finds at tube-sample index n (n = ring position - pad) of a voice whose ring received `total` samples
// (the flush's zeros included): past the last one the ring still holds the sample written one lap earlier.
TRM_HD long long src_ring_sample(long long n, long long total)
{
    while (n >= total) n -= (long long)kSrcRing;
    return n;
}

TRM_HD uint32_t src_position(uint32_t k, uint32_t inc) { return (uint32_t)(((uint64_t)k * inc) >> 16); }

}  // namespace trm
