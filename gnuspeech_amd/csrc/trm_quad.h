// trm_quad.h -- the same model as trm_lane.h, re-indexed for SMALL batches (trm_quad.hip): a voice owns
// FOUR lanes instead of one.
//   * In the tube stage the four lanes are four "parts" of the tube: each holds a contiguous run of
//     scattering junctions (below), all stepped at once; junction values that cross a part boundary move
//     by DPP (a row rotation by one bank of four lanes with a bank mask).
//   * In the feed-forward stages (oscillator, FIR + mixing, coefficients) the four lanes are four
//     consecutive tube samples ("slots"); only the oscillator phase is a recurrence there, and it is a
//     prefix sum over the four slots.
// The arithmetic per value is that of trm_lane.h (same operations in the same order), so the host model
// (F = Q4, tests/_emul) reproduces tube_step bit for bit; the functions are templates over the value type
// F: float on the device (one lane = one part), Q4 on the host (four parts side by side).
#pragma once

#include "trm_lane.h"

namespace trm {

constexpr int kPart0 = 1, kPart1 = 2, kPart2 = 4, kPart3 = 8, kPartAll = 15;

// ---------------------------------------------------------------- host model of one voice's four lanes
struct Q4 {
    float v[4];
    Q4() = default;
    explicit Q4(float x) { v[0] = v[1] = v[2] = v[3] = x; }
    Q4(float a, float b, float c, float d) { v[0] = a; v[1] = b; v[2] = c; v[3] = d; }
};
inline Q4 operator+(Q4 a, Q4 b) { return Q4(a.v[0] + b.v[0], a.v[1] + b.v[1], a.v[2] + b.v[2], a.v[3] + b.v[3]); }
inline Q4 operator-(Q4 a, Q4 b) { return Q4(a.v[0] - b.v[0], a.v[1] - b.v[1], a.v[2] - b.v[2], a.v[3] - b.v[3]); }
inline Q4 operator*(Q4 a, Q4 b) { return Q4(a.v[0] * b.v[0], a.v[1] * b.v[1], a.v[2] * b.v[2], a.v[3] * b.v[3]); }
inline Q4 operator-(Q4 a) { return Q4(-a.v[0], -a.v[1], -a.v[2], -a.v[3]); }
inline Q4 fma_f(Q4 a, Q4 b, Q4 c)
{
    return Q4(fma_f(a.v[0], b.v[0], c.v[0]), fma_f(a.v[1], b.v[1], c.v[1]), fma_f(a.v[2], b.v[2], c.v[2]),
              fma_f(a.v[3], b.v[3], c.v[3]));
}
// parts in MASK receive src of part (p - K) mod 4, the others keep old
template <int K, int MASK>
inline Q4 q_take(Q4 old, Q4 src)
{
    Q4 r = old;
    for (int p = 0; p < 4; p++)
        if ((MASK >> p) & 1) r.v[p] = src.v[(p - K) & 3];
    return r;
}

#if defined(__HIP__)
// Lanes of a row of 16: bank b = lanes 4b..4b+3 = part b of four voices.  row_ror:4K hands every lane the
// value of the lane 4K below it in its row, i.e. of part (p - K) mod 4 of the same voice; the bank mask
// picks the parts that take it (checked on the hardware by tools/ubench/dpp_check.hip).
template <int K, int MASK>
__device__ __forceinline__ float q_take(float old, float src)
{
    constexpr int ctrl = K == 0 ? 0xE4 /* quad_perm:[0,1,2,3] */ : 0x120 + 4 * K /* row_ror:4K */;
    if (MASK == kPartAll)       // every part takes: no lane keeps `old`, so the destination need not start as a copy of it
        return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, src), ctrl, 0xF, 0xF, false));
    return __builtin_bit_cast(
        float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, src), ctrl, 0xF, MASK, false));
}
template <int K, int MASK>
__device__ __forceinline__ double q_take(double old, double src)
{
    const unsigned long long o = __builtin_bit_cast(unsigned long long, old), s = __builtin_bit_cast(unsigned long long, src);
    constexpr int ctrl = K == 0 ? 0xE4 : 0x120 + 4 * K;
    const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp((int)(unsigned)o, (int)(unsigned)s, ctrl, 0xF, MASK, false);
    const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp((int)(unsigned)(o >> 32), (int)(unsigned)(s >> 32), ctrl, 0xF, MASK, false);
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
#endif

// ================================================================ tube stage: four parts per voice
// Junction Jn sits between sections Sn and Sn+1 and produces T = oT[n] (the top wave entering Sn+1) and
// B = oB[n-1] (the bottom wave entering Sn) from a = oT[n-1] and b = oB[n] (TRMTubeModel.m:778-853);
// nasal junction Nn likewise over nT/nB.  Four generic "rounds" r = 0..3 per part:
//   part 0:  J1  J2  J3  --      + glottis end
//   part 1:  J5  J6  J7  --      + the three-way junction J4;  J6 is the junction-less S6|S7 boundary (k = 0)
//   part 2:  J8  J9  --  N5      + BOTH ends (mouth after J9, nose after N5) as one two-wide filter, throat
//   part 3:  N1  N2  N3  N4
// The frication band-pass runs in every part (all of them inject its output).  Rounds are stepped two at a
// time as two-wide values (v_pk_* on the device): rounds (0, 2) and rounds (1, 3), because round r+1 reads
// round r's T and round r reads round r+1's B: with this pairing the a-inputs of rounds (1, 3) ARE the
// (T0, T2) pair, the b-inputs of rounds (0, 2) are the (B1, B3) pair, and in part 2 the two ends read the
// (T1, T3) pair and write the b-inputs of rounds (1, 3).
struct PartRecord {
    float kk[4];        // junction transmission factors (1 + k) d of rounds 0, 2, 1, 3 (tube_step's working form; part 1's
                        // idle round 3: alphaU; other idle rounds 0; the junction-less S6|S7 boundary: d)
    float tp[4];        // frication taps of rounds 0, 2, 1, 3          (part 1's idle round 3: FC3)
};

TRM_HD void pack_part_kk(const Coefs &K, const Const &C, PartRecord R[4])      // needs coef_sample_area's fields
{
    const float kk[4][4] = {{K.td[0], K.td[2], K.td[1], 0.0f}, {K.td[3], K.td[4], C.damping, K.alphaU},
                            {K.td[5], 0.0f, K.td[6], C.nasalTd[3]}, {K.ntd1, C.nasalTd[1], C.nasalTd[0], C.nasalTd[2]}};
    for (int p = 0; p < 4; p++)
        for (int i = 0; i < 4; i++) R[p].kk[i] = kk[p][i];
}
TRM_HD void pack_part_tp(const Coefs &K, PartRecord R[4])                       // needs coef_sample_fric's fields
{
    const float tp[4][4] = {{0.0f, K.tap[1], K.tap[0], 0.0f}, {K.tap[3], K.tap[5], K.tap[4], K.tap[2]},
                            {K.tap[6], 0.0f, K.tap[7], 0.0f}, {0.0f, 0.0f, 0.0f, 0.0f}};
    for (int p = 0; p < 4; p++)
        for (int i = 0; i < 4; i++) R[p].tp[i] = tp[p][i];
}
TRM_HD void pack_part_records(const Coefs &K, const Const &C, PartRecord R[4])
{
    pack_part_kk(K, C, R);
    pack_part_tp(K, R);
}

// Per-sample coefficients every part reads: the band-pass DOUBLED (y = 2 (alpha (x - x2) + gamma y1 - beta y2),
// TRMFilters.m:19-29, with the factor folded in: exact in binary floating point) and the end filters'
// {C8 a10, NC6 a10} (the reflection filters' input gains), {1 + C8, 1 + NC6} (:820-836, :848-852).
struct SharedRecord {
    float bpA2, bpB2, bpG2, pad_;
    float endK[2], endOnePlus[2];
};

TRM_HD void pack_shared_bp(const Coefs &K, SharedRecord &R)                     // coef_sample_fric's fields
{
    R.bpA2 = 2.0f * K.bpAlpha; R.bpB2 = 2.0f * K.bpBeta; R.bpG2 = 2.0f * K.bpGamma; R.pad_ = 0.0f;
}
TRM_HD void pack_shared_end(const Coefs &K, const Const &C, SharedRecord &R)    // coef_sample_area's fields
{
    R.endK[0] = K.k8a;
    R.endK[1] = C.nasalK6a;
    R.endOnePlus[0] = K.onePlusK8;
    R.endOnePlus[1] = C.onePlusNK6;
}
TRM_HD void pack_shared_record(const Coefs &K, const Const &C, SharedRecord &R)
{
    pack_shared_bp(K, R);
    pack_shared_end(K, C, R);
}

// two-wide values
struct Q4P {
    Q4 x, y;
};
inline Q4P operator+(Q4P a, Q4P b) { return Q4P{a.x + b.x, a.y + b.y}; }
inline Q4P operator-(Q4P a, Q4P b) { return Q4P{a.x - b.x, a.y - b.y}; }
inline Q4P operator*(Q4P a, Q4P b) { return Q4P{a.x * b.x, a.y * b.y}; }
inline Q4P pk_make(Q4 x, Q4 y) { return Q4P{x, y}; }
inline Q4P pk_fma(Q4P a, Q4P b, Q4P c) { return Q4P{fma_f(a.x, b.x, c.x), fma_f(a.y, b.y, c.y)}; }
template <class F> struct PairOf;
template <> struct PairOf<Q4> { typedef Q4P type; };
#if defined(__HIP__)
typedef float v2f_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2f_t pk_make(float x, float y) { return v2f_t{x, y}; }
__device__ __forceinline__ v2f_t pk_fma(v2f_t a, v2f_t b, v2f_t c) { return __builtin_elementwise_fma(a, b, c); }
template <> struct PairOf<float> { typedef v2f_t type; };
#endif

template <class F>
struct QuadState {
    typedef typename PairOf<F>::type P;
    P TA, TB;               // generic junction outputs {T0, T2}, {T1, T3}
    P BA, BB;               //                          {B0, B2}, {B1, B3}
    F A0;                   // part 0: oT[0], the glottis end (= round 0's a-input)
    F jT, jB, jN;           // part 1: the three-way junction's oT[4], oB[3], nT[0]
    P eB;                   // part 2: {oB[9], nB[5]}: mouth and nose reflections (= rounds (1, 3)'s b-inputs)
    P reflY, radX, radY;    // end filter memories {mouth, nose}
    F thY;                  // throat low-pass memory
    F bx1, bx2, by1, by2;   // frication band-pass memory
};

template <class F>
TRM_HD void quad_reset(QuadState<F> &S)
{
    const F z = F(0.0f);
    const typename PairOf<F>::type zz = pk_make(z, z);
    S.TA = zz; S.TB = zz; S.BA = zz; S.BB = zz;
    S.A0 = S.jT = S.jB = S.jN = z;
    S.eB = zz; S.reflY = zz; S.radX = zz; S.radY = zz;
    S.thY = z;
    S.bx1 = S.bx2 = S.by1 = S.by2 = z;
}

// The two recurrences that only FEED the tube (nothing of the tube's state enters them), as single
// evaluations: on the device they run in feed-forward waves, serially over the four time slots of a voice.
//   frication band-pass (TRMFilters.m:19-29), coefficients doubled (see SharedRecord)
template <class F>
TRM_HD F bandpass_eval(F bpA2, F bpB2, F bpG2, F x, F x2, F y1, F y2)
{
    return fma_f(bpA2, x - x2, fma_f(bpG2, y1, -(bpB2 * y2)));
}
//   throat low-pass (:341, TRMFilters.m:72-77)
template <class F>
TRM_HD F throat_eval(const Const &C, F thr, F y1)
{
    return fma_f(F(C.ta0), thr, F(C.tb1) * y1);      // (the FMA written out: trm_lane.h tube_step)
}

// One tube sample given the throat output `ty` of that sample and the frication INJECTIONS inA / inB = this part's
// taps x the band-pass output of that sample (rounds (0, 2) and (1, 3); part 1's idle round 3: the three-way junction's).
// Returns the tube-rate output in PART 2 (other parts: unspecified).
template <class F>
TRM_HD F tube_quad_core(QuadState<F> &S, const Const &C, F gin, F ty, typename PairOf<F>::type endK,
                        typename PairOf<F>::type endOnePlus, typename PairOf<F>::type kA, typename PairOf<F>::type kB,
                        typename PairOf<F>::type inA, typename PairOf<F>::type inB)
{
    typedef typename PairOf<F>::type P;
    const F d = F(C.damping);

    // ---- gather every junction's two inputs from the previous sample's outputs
    F a0 = S.A0;                                   // J1 <- glottis end
    a0 = q_take<0, kPart1>(a0, S.jT);              // J5 <- three-way oT[4]
    a0 = q_take<1, kPart2>(a0, S.TA.y);            // J8 <- J7.T (part 1 round 2)
    a0 = q_take<2, kPart3>(a0, S.jN);              // N1 <- three-way nT[0] (part 1)
    const F a3 = q_take<3, kPart2>(S.TA.y, S.TB.y);    // round 3 <- own T2; N5 <- N4.T (part 3 round 3)
    F b2 = S.BB.y;                                 // N3 <- N4.B
    b2 = q_take<3, kPart0>(b2, S.jB);              // J3 <- three-way oB[3] (part 1)
    b2 = q_take<3, kPart1>(b2, S.BA.x);            // J7 <- J8.B (part 2 round 0)
    const F b1 = q_take<0, kPart0 | kPart1 | kPart3>(S.eB.x, S.BA.y);  // round 1 <- own B2; J9 <- mouth reflection oB[9]
    const F b3 = q_take<1, kPart3>(S.eB.y, S.BB.y);    // N5 <- nose reflection nB[5]; N4 <- N5.B (part 2 round 3)
    const F x1 = q_take<1, kPartAll>(S.TA.y, S.TA.y);  // three-way: oT[3] = J3.T (part 0 round 2)
    const F x2 = S.BA.x;                               //            oB[4] = J5.B; also the glottis end's oB[0] = J1.B
    const F x3 = q_take<2, kPartAll>(S.BA.x, S.BA.x);  //            nB[0] = N1.B (part 3 round 0)
    const P ei = S.TB;                                 // ends (part 2): mouth oT[9] = J9.T, nose nT[5] = N5.T
    const P aA = pk_make(a0, S.TB.x), aB = pk_make(S.TA.x, a3);    // a2 = T1, a1 = T0
    const P bA = pk_make(S.BB.x, b2), bB = pk_make(b1, b3);        // b0 = B1

    // ---- generic junctions (:783-816, :838-846), two rounds per operation
    const P dd = pk_make(d, d);
    // T = d b + t (a - b) + tap fr, B = T - d (a - b): kA / kB carry t = (1 + k) d (tube_step)
    const P dfA = aA - bA, dfB = aB - bB;
    const P ndd = pk_make(-d, -d);
    const P TA = pk_fma(kA, dfA, dd * bA), TB = pk_fma(kB, dfB, dd * bB);
    S.BA = pk_fma(ndd, dfA, TA);
    S.BB = pk_fma(ndd, dfB, TB);
    S.TA = TA + inA;
    S.TB = TB + inB;
    // ---- glottis end (:781)
    S.A0 = x2 * d + gin;
    // ---- three-way junction (:801-806); the three alphas sum to 2 (:733-736)
    {
        const F aU = kB.y, aLR = fma_f(F(-0.5f), aU, F(1.0f));
        F jp = fma_f(aLR, x1, fma_f(aLR, x2, aU * x3));
        S.jB = (jp - x1) * d;
        S.jT = (jp - x2) * d + inB.y;
        S.jN = (jp - x3) * d;
    }
    // ---- mouth and nose ends: reflection + radiation (:820-836, :848-852, TRMFilters.m:47-60), two-wide
    const P cf = pk_make(F(C.mCoeff), F(C.nCoeff));
    P refl = pk_fma(endK, ei, cf * S.reflY);
    S.reflY = refl;
    S.eB = dd * refl;
    P rin = endOnePlus * ei;
    P rad = cf * (rin - S.radX + S.radY);
    S.radX = rin; S.radY = rad;
    // ---- the output sum (:346): mouth + nose, then the throat
    F out = rad.x + rad.y;
    return ty * F(C.throatGain) + out;
}

// The whole sample in one call (host model): band-pass and throat state live in S.
template <class F>
TRM_HD F tube_quad_step(QuadState<F> &S, const Const &C, F gin, F sig, F thr, F bpA2, F bpB2, F bpG2,
                        typename PairOf<F>::type endK, typename PairOf<F>::type endOnePlus, typename PairOf<F>::type kA,
                        typename PairOf<F>::type kB, typename PairOf<F>::type tA, typename PairOf<F>::type tB)
{
    const F fr = bandpass_eval(bpA2, bpB2, bpG2, sig, S.bx2, S.by1, S.by2);
    S.bx2 = S.bx1; S.bx1 = sig; S.by2 = S.by1; S.by1 = fr;
    const F ty = throat_eval(C, thr, S.thY);
    S.thY = ty;
    const typename PairOf<F>::type ff = pk_make(fr, fr);
    return tube_quad_core(S, C, gin, ty, endK, endOnePlus, kA, kB, tA * ff, tB * ff);
}

// ================================================================ oscillator, time-slot form
// Tracks evaluated at the lane's own sample and advanced kSlots samples per block.
constexpr int kSlots = 4;

struct OscSlotTrack {
    double f0, f0Step;          // f0 at this lane's sample; (per-sample ratio)^kSlots
    double axGeo, axStep;
    double glot0, glotDelta;    // dB = glot0 + j * glotDelta
    float aspBase, aspDelta;
};

// `j` = the lane's position in the new control period, 0 <= j < 2^kLog2Slots; the lane's tracks then advance
// 2^kLog2Slots samples at a time (four: trm_quad.hip; eight: trm_oct.hip).
// The set-up's four exponentials (frequency(), TRMUtility.m:44-47; amplitude(), :26-41) are 2^x of these arguments:
// {pitch at the period's start, pitch step per sample, glottal volume at the start, its step per sample}
TRM_HD void osc_slot_exp_args(const Const &C, const float *prev, const float *cur, double x[4])
{
    const double kLog2_10_over_20 = 0.16609640474436813;
    const double p0 = (double)prev[0], dp = ((double)cur[0] - p0) * C.invControlPeriodD;
    const double v0 = (double)prev[1], dv = ((double)cur[1] - v0) / (double)C.controlPeriod;
    x[0] = (p0 + 3.0) * (1.0 / 12.0);
    x[1] = dp * (1.0 / 12.0);
    x[2] = (v0 - 60.0) * kLog2_10_over_20;
    x[3] = dv * kLog2_10_over_20;
}
// ... and the track from their values e[k] = 2^x[k]
template <int kLog2Slots>
TRM_HD void osc_slot_from_exps(OscSlotTrack &T, const Const &C, const float *prev, const float *cur, int j, const double e[4])
{
    const double v0 = (double)prev[1], dv = ((double)cur[1] - v0) / (double)C.controlPeriod;
    double f0 = 220.0 * e[0], r = e[1], ax = e[2], q = e[3];
    // r^j, q^j by the bits of j; r^(2^kLog2Slots) is what is left in r
    double rj = 1.0, qj = 1.0;
    for (int b = 0; b < kLog2Slots; b++) {
        rj *= ((j >> b) & 1) ? r : 1.0;
        qj *= ((j >> b) & 1) ? q : 1.0;
        r *= r;
        q *= q;
    }
    T.f0 = f0 * rj;
    T.f0Step = r;
    T.axGeo = ax * qj;
    T.axStep = q;
    T.glot0 = v0;
    T.glotDelta = dv;
    T.aspBase = prev[2];
    T.aspDelta = (cur[2] - prev[2]) * C.invControlPeriod;
}
template <int kLog2Slots>
TRM_HD void osc_slot_setup_pow2(OscSlotTrack &T, const Const &C, const float *prev, const float *cur, int j)
{
    double x[4], e[4];
    osc_slot_exp_args(C, prev, cur, x);
    for (int k = 0; k < 4; k++) e[k] = exp2_d(x[k]);
    osc_slot_from_exps<kLog2Slots>(T, C, prev, cur, j, e);
}
TRM_HD void osc_slot_setup(OscSlotTrack &T, const Const &C, const float *prev, const float *cur, int j)
{
    osc_slot_setup_pow2<2>(T, C, prev, cur, j);
}

// The representative of x (> -1) in (-1, 511]: what repeated `pos > 511 ? pos - 512 : pos` arrives at
// (TRMWavetable.m:28-34,165-168).
TRM_HD double osc_wrap(double x) { return x - 512.0 * __builtin_ceil((x - 511.0) * (1.0 / 512.0)); }

// ================================================================ 49-tap FIR, direct form over a window
// y[m] = sum_{k=0..24} c[2k] b[m-k] + sum_{k=0..23} c[2k+1] a[m-k] (TRMFIRFilter.m:116-146, decimating by 2:
// a, b = the two oversampled oscillator reads of a tube sample).  The window holds 26 samples starting at
// the even index s0 = m - 24 - o, o = m & 1, so that it can be read as aligned 16-byte pairs of (a, b);
// window slot i is sample m - k with k = 24 + o - i, and taps outside 0..24 are zero.
constexpr int kFirWin = 26;

TRM_HD float fir_window_tap_b(const float *fir, int o, int i)
{
    int k = 24 + o - i;
    int t = 2 * k;
    return (k >= 0 && k <= 24) ? fir[t < kFirUnique ? t : (kFirTaps - 1) - t] : 0.0f;
}
TRM_HD float fir_window_tap_a(const float *fir, int o, int i)
{
    int k = 24 + o - i;
    int t = 2 * k + 1;
    return (k >= 0 && k <= 23) ? fir[t < kFirUnique ? t : (kFirTaps - 1) - t] : 0.0f;
}

// win = 26 x (a, b) interleaved; ca/cb = the window taps of this lane's parity.  Four partial sums.
TRM_HD float fir_direct(const float *win, const float *ca, const float *cb)
{
    float s0 = win[0] * ca[0], s1 = win[1] * cb[0], s2 = win[2] * ca[1], s3 = win[3] * cb[1];
    for (int i = 2; i < kFirWin; i += 2) {
        s0 = fma_f(win[2 * i], ca[i], s0);
        s1 = fma_f(win[2 * i + 1], cb[i], s1);
        s2 = fma_f(win[2 * i + 2], ca[i + 1], s2);
        s3 = fma_f(win[2 * i + 3], cb[i + 1], s3);
    }
    return (s0 + s2) + (s1 + s3);
}

}  // namespace trm
