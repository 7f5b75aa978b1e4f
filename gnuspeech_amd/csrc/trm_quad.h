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
//   part 0:  J1  J2  J3  --      + glottis end, nose end (reflection + radiation)
//   part 1:  J5  J6  J7  --      + the three-way junction J4;  J6 is the junction-less S6|S7 boundary (k = 0)
//   part 2:  J8  J9  N5  --      + mouth end, throat
//   part 3:  N1  N2  N3  N4
// The frication band-pass runs in every part (all of them inject its output).
struct PartRecord {
    float kk[4];        // round r's scattering coefficient; slot 3 of parts 1 / 2: alphaU / 1 + C8
    float tp[4];        // round r's frication tap;          slot 3 of part 1: FC3 (the junction's tap)
};

TRM_HD void pack_part_records(const Coefs &K, const Const &C, PartRecord R[4])
{
    R[0] = PartRecord{{K.k[0], K.k[1], K.k[2], 0.0f}, {0.0f, K.tap[0], K.tap[1], 0.0f}};
    R[1] = PartRecord{{K.k[3], 0.0f, K.k[4], K.alphaU}, {K.tap[3], K.tap[4], K.tap[5], K.tap[2]}};
    R[2] = PartRecord{{K.k[5], K.k[6], C.nasalK[3], K.onePlusK8}, {K.tap[6], K.tap[7], 0.0f, 0.0f}};
    R[3] = PartRecord{{K.nk1, C.nasalK[0], C.nasalK[1], C.nasalK[2]}, {0.0f, 0.0f, 0.0f, 0.0f}};
}

template <class F>
struct QuadState {
    F T[4], B[4];           // outputs of this part's generic junctions
    F G;                    // part 0: oT[0], the glottis end
    F jT, jB, jN;           // part 1: the three-way junction's oT[4], oB[3], nT[0]
    F eB;                   // part 2: oB[9] (mouth reflection); part 0: nB[5] (nose reflection)
    F reflY, radX, radY;    // end filter memories (part 2 mouth, part 0 nose)
    F thY;                  // throat low-pass memory
    F bx1, bx2, by1, by2;   // frication band-pass memory
};

template <class F>
struct QuadConst {          // per-part constants of the end filters: part 0 nose, part 2 mouth
    F endCoeff, endA10, endK, endOnePlus;
};

template <class F>
TRM_HD void quad_reset(QuadState<F> &S)
{
    for (int i = 0; i < 4; i++) { S.T[i] = F(0.0f); S.B[i] = F(0.0f); }
    S.G = S.jT = S.jB = S.jN = S.eB = F(0.0f);
    S.reflY = S.radX = S.radY = S.thY = F(0.0f);
    S.bx1 = S.bx2 = S.by1 = S.by2 = F(0.0f);
}

// One tube sample.  Returns the tube-rate output in PART 2 (other parts: unspecified).
template <class F>
TRM_HD F tube_quad_step(QuadState<F> &S, const QuadConst<F> &Q, const Const &C, F gin, F sig, F thr, F bpAlpha, F bpBeta,
                        F bpGamma, const F *kk, const F *tp)
{
    const F d = F(C.damping);
    // frication band-pass (TRMFilters.m:19-29)
    F fr = F(2.0f) * fma_f(bpAlpha, sig - S.bx2, fma_f(bpGamma, S.by1, -(bpBeta * S.by2)));
    S.bx2 = S.bx1; S.bx1 = sig; S.by2 = S.by1; S.by1 = fr;

    // ---- gather every junction's two inputs from the previous sample's outputs
    F a0 = S.G;                                   // J1 <- glottis end
    a0 = q_take<0, kPart1>(a0, S.jT);             // J5 <- three-way oT[4]
    a0 = q_take<1, kPart2>(a0, S.T[2]);           // J8 <- J7.T (part 1)
    a0 = q_take<2, kPart3>(a0, S.jN);             // N1 <- three-way nT[0] (part 1)
    const F b0 = S.B[1];
    const F a1 = S.T[0];
    const F b1 = q_take<0, kPart2>(S.B[2], S.eB); // J9 <- mouth reflection oB[9]
    const F a2 = q_take<3, kPart2>(S.T[1], S.T[3]);   // N5 <- N4.T (part 3)
    F b2 = S.B[3];                                // N3 <- N4.B
    b2 = q_take<3, kPart0>(b2, S.jB);             // J3 <- three-way oB[3] (part 1)
    b2 = q_take<3, kPart1>(b2, S.B[0]);           // J7 <- J8.B (part 2)
    b2 = q_take<2, kPart2>(b2, S.eB);             // N5 <- nose reflection nB[5] (part 0)
    const F a3 = S.T[2];
    const F b3 = q_take<1, kPart3>(S.B[3], S.B[2]);   // N4 <- N5.B (part 2); other parts idle in round 3
    const F x1 = q_take<1, kPartAll>(S.T[2], S.T[2]); // three-way: oT[3] = J3.T (part 0)
    const F x2 = S.B[0];                              //            oB[4] = J5.B
    const F x3 = q_take<2, kPartAll>(S.B[0], S.B[0]); //            nB[0] = N1.B (part 3)
    const F ei = q_take<2, kPart0>(S.T[1], S.T[2]);   // ends: mouth oT[9] = J9.T; nose nT[5] = N5.T (part 2)

    // ---- generic junctions (:783-816, :838-846)
    const F as[4] = {a0, a1, a2, a3}, bs[4] = {b0, b1, b2, b3};
    for (int r = 0; r < 4; r++) {
        F dl = kk[r] * (as[r] - bs[r]);
        S.T[r] = (as[r] + dl) * d + tp[r] * fr;
        S.B[r] = (bs[r] + dl) * d;
    }
    // ---- glottis end (:781)
    S.G = x2 * d + gin;
    // ---- three-way junction (:801-806); the three alphas sum to 2 (:733-736)
    {
        const F aU = kk[3], aLR = fma_f(F(-0.5f), aU, F(1.0f));
        F jp = aLR * x1 + (aLR * x2 + aU * x3);
        S.jB = (jp - x1) * d;
        S.jT = (jp - x2) * d + tp[3] * fr;
        S.jN = (jp - x3) * d;
    }
    // ---- mouth / nose ends: reflection + radiation (:820-836, :848-852, TRMFilters.m:47-60)
    const F onePlus = q_take<0, kPart2>(Q.endOnePlus, kk[3]);
    const F kend = q_take<0, kPart2>(Q.endK, kk[3] - F(1.0f));   // C8 = (1 + C8) - 1
    F refl = Q.endA10 * (kend * ei) + Q.endCoeff * S.reflY;
    S.reflY = refl;
    S.eB = d * refl;
    F rin = onePlus * ei;
    F rad = Q.endCoeff * (rin - S.radX + S.radY);
    S.radX = rin; S.radY = rad;
    // ---- throat (:341, TRMFilters.m:72-77) and the output sum, in part 2
    F ty = F(C.ta0) * thr + F(C.tb1) * S.thY;
    S.thY = ty;
    F out = rad + q_take<2, kPartAll>(rad, rad);     // mouth + nose (part 0)
    return ty * F(C.throatGain) + out;
}

// End-filter constants by part.
TRM_HD void quad_const_parts(const Const &C, float endCoeff[4], float endA10[4], float endK[4], float endOnePlus[4])
{
    for (int p = 0; p < 4; p++) { endCoeff[p] = C.mCoeff; endA10[p] = C.mA10; endK[p] = 0.0f; endOnePlus[p] = 1.0f; }
    endCoeff[0] = C.nCoeff; endA10[0] = C.nA10; endK[0] = C.nasalK[4]; endOnePlus[0] = C.onePlusNK6;
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

// `j` = the lane's position in the new control period, 0 <= j < kSlots.
TRM_HD void osc_slot_setup(OscSlotTrack &T, const Const &C, const float *prev, const float *cur, int j)
{
    const double kLog2_10_over_20 = 0.16609640474436813;
    double p0 = (double)prev[0], dp = ((double)cur[0] - p0) * C.invControlPeriodD;
    double f0 = 220.0 * exp2_d((p0 + 3.0) * (1.0 / 12.0));
    double r = exp2_d(dp * (1.0 / 12.0));
    double v0 = (double)prev[1], dv = ((double)cur[1] - v0) / (double)C.controlPeriod;
    double ax = exp2_d((v0 - 60.0) * kLog2_10_over_20);
    double q = exp2_d(dv * kLog2_10_over_20);
    double r2 = r * r, q2 = q * q;
    double rj = ((j & 1) ? r : 1.0) * ((j & 2) ? r2 : 1.0);
    double qj = ((j & 1) ? q : 1.0) * ((j & 2) ? q2 : 1.0);
    T.f0 = f0 * rj;
    T.f0Step = r2 * r2;
    T.axGeo = ax * qj;
    T.axStep = q2 * q2;
    T.glot0 = v0;
    T.glotDelta = dv;
    T.aspBase = prev[2];
    T.aspDelta = (cur[2] - prev[2]) * C.invControlPeriod;
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
