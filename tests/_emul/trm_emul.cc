// trm_emul.cc -- TEST INFRASTRUCTURE.  Runs gnuspeech_amd/csrc/trm_lane.h (the exact per-lane
// stage arithmetic the HIP kernel executes) serially on the host, one voice at a time, so the
// fp32/fp64 precision plan can be checked against the oracle in a container without a GPU.  Never
// linked into libtrm_hip.so and never used by the product path.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "../../gnuspeech_amd/csrc/trm_lane.h"
#include "../../gnuspeech_amd/csrc/trm_oct.h"
#include "../../gnuspeech_amd/csrc/trm_setup.h"

using namespace trm;

// tract: the loop order of Applications/TRAcT/tube.c as the streaming kernel runs it in TRM_STREAM_MODE_TRACT (frame f held
// for control period f, frication amplitude x10 in the coefficient stage, x100 on the converter's output)
static int emul_synthesize(const trm_input_params *p, const float *frames, size_t nframes,
                           float *out, size_t cap, uint32_t *nout, float *maxv, float *tube, bool tract)
{
    Const C;
    trm_derived d;
    int rc = build_const(*p, C, d);
    if (rc) return rc;
    if (tract) C.fricGain = 10.0f;
    if (!C.upsample) return TRM_ERANGE;
    static std::vector<float> rows, sine;
    if (rows.empty()) { build_src_rows(rows); build_sine_table(sine); }
    *nout = 0; *maxv = 0.f;
    if (nframes == 0) return TRM_OK;
    size_t ntube = (nframes - 1) * (size_t)C.controlPeriod;
    std::vector<float> lp(ntube + 1);
    {   // TRMUtility.m:71-85 + TRMFilters.m:81-86 in fp64 (the device does this in trm_noise_kernel)
        double seed = 0.7892347, x1 = 0.0;
        for (size_t i = 0; i < ntube; i++) {
            double prod = seed * 377.0;
            seed = prod - (int)prod;
            double nz = seed - 0.5;
            lp[i] = (float)(nz + x1);
            x1 = nz;
        }
    }
    ExciteState ES; ExciteTrack ET; CoefTrack CT; TubeState TS;
    excite_reset(ES); tube_reset(TS);
    auto sineLookup = [&](int i) { return sine_table(i); };
    // tube-rate signal, with the converter's 25 zeros of pre-roll and 2*pad zeros of flush around it
    std::vector<float> sig(25 + ntube + 2 * C.padSize, 0.0f);
    size_t n = 0;
    for (size_t f = 1; f < nframes; f++) {
        excite_track_setup(ET, C, frames + 16 * (tract ? f : f - 1), frames + 16 * f);
        coef_track_setup(CT, C, frames + 16 * (tract ? f : f - 1), frames + 16 * f);
        for (int j = 0; j < C.controlPeriod; j++) {
            Excitation E = excite_sample(ES, ET, C, C.fir, j, lp[n], sineLookup);
            Coefs K = tract ? coef_sample<true>(CT, C, j) : coef_sample(CT, C, j);
            float s = tube_sample(TS, C, E, K);
            if (tube) tube[n] = s;
            sig[25 + n] = s;
            n++;
        }
    }
    uint64_t total = count_outputs(d, ntube);
    float mx = 0.f;
    for (uint64_t k = 0; k < total; k++) {
        uint32_t ph = src_phase((uint32_t)k, C.timeRegisterIncrement);
        uint32_t e = src_position((uint32_t)k, C.timeRegisterIncrement);
        float y = src_dot(&sig[e], &rows[(size_t)ph * kSrcRowC]);
        if (tract) y *= 100.0f;
        if (k < cap) out[k] = y;
        float a = fabsf(y);
        if (a > mx) mx = a;
    }
    *nout = (uint32_t)total;
    *maxv = mx;
    return TRM_OK;
}

extern "C" int trm_emul_synthesize(const trm_input_params *p, const float *frames, size_t nframes,
                                   float *out, size_t cap, uint32_t *nout, float *maxv, float *tube)
{
    return emul_synthesize(p, frames, nframes, out, cap, nout, maxv, tube, false);
}

extern "C" int trm_emul_synthesize_tract(const trm_input_params *p, const float *frames, size_t nframes,
                                         float *out, size_t cap, uint32_t *nout, float *maxv, float *tube)
{
    return emul_synthesize(p, frames, nframes, out, cap, nout, maxv, tube, true);
}

// ---------------------------------------------------------------------------------------------
// Time-split synthesis (the host model of trm_kernels.hip's segment instance): the utterance is cut every `segPeriods`
// control periods (the first segment is segPeriods + warmPeriods long); every segment after the first starts from REST `warmPeriods` control periods early -- only the
// oscillator position is the true one (an exact prefix sum, osc_increment) and the noise sequence is addressed by its
// index -- and the converter outputs whose read position lies in the segment proper are taken from it.  What the warm-up
// forgets decays like the tube's slowest pole (damping^n for a closed tract); tests/test_quad_model.py and
// tools/timesplit_study.py measure it against the oracle.
extern "C" int trm_emul_synthesize_split(const trm_input_params *p, const float *frames, size_t nframes,
                                         float *out, size_t cap, uint32_t *nout, float *maxv, uint32_t segPeriods, uint32_t warmPeriods)
{
    Const C;
    trm_derived d;
    int rc = build_const(*p, C, d);
    if (rc) return rc;
    if (!C.upsample || segPeriods == 0) return TRM_ERANGE;
    static std::vector<float> rows;
    if (rows.empty()) build_src_rows(rows);
    *nout = 0; *maxv = 0.f;
    if (nframes == 0) return TRM_OK;
    const size_t CP = (size_t)C.controlPeriod, nper = nframes - 1, ntube = nper * CP;
    const uint32_t inc = C.timeRegisterIncrement;
    std::vector<float> lp(ntube + 1);
    {
        double seed = 0.7892347, x1 = 0.0;
        for (size_t i = 0; i < ntube; i++) {
            double prod = seed * 377.0;
            seed = prod - (int)prod;
            double nz = seed - 0.5;
            lp[i] = (float)(nz + x1);
            x1 = nz;
        }
    }
    auto sineLookup = [&](int i) { return sine_table(i); };
    auto wrap = [](double v) { return v > 511.0 ? v - 512.0 : v; };
    // oscillator position at the start of every control period (what the device's prefix pass computes)
    std::vector<double> phase(nper + 1, 0.0);
    {
        ExciteTrack T;
        double pos = 0.0;
        for (size_t f = 1; f <= nper; f++) {
            excite_track_setup(T, C, frames + 16 * (f - 1), frames + 16 * f);
            for (size_t j = 0; j < CP; j++) {
                const double i2 = osc_increment(T.f0, C);
                pos = wrap(wrap(pos + i2) + i2);
                T.f0 *= T.f0Ratio;
            }
            phase[f] = pos;
        }
    }
    const uint64_t total = count_outputs(d, ntube);
    auto outputs_through = [&](uint64_t endSample) { return endSample == 0 ? 0ull : ((endSample << 16) - 1) / inc + 1; };
    float mx = 0.f;
    // segment boundaries as the library lays them (trm_capi.cc split_segments): the first segment is segPeriods + warmPeriods long
    auto seg_begin = [&](size_t sg) { return sg == 0 ? (size_t)0 : (size_t)segPeriods + warmPeriods + (sg - 1) * segPeriods; };
    size_t nseg = 1;
    while (seg_begin(nseg) < nper) nseg++;
    for (size_t sg = 0; sg < nseg; sg++) {
        const size_t pLo = seg_begin(sg), pHi = seg_begin(sg + 1) < nper ? seg_begin(sg + 1) : nper;
        const size_t pStart = pLo > warmPeriods ? pLo - warmPeriods : 0;
        const size_t nBase = pStart * CP, nLocal = (pHi - pStart) * CP;
        const bool last = sg + 1 == nseg;
        // local tube-rate signal: 25 positions of pre-roll (zeros), the samples, the flush zeros
        std::vector<float> sig(25 + nLocal + 2 * C.padSize + 8, 0.0f);
        ExciteState ES; ExciteTrack ET; CoefTrack CT; TubeState TS;
        excite_reset(ES); tube_reset(TS);
        ES.oscPos = phase[pStart];
        size_t n = 0;
        for (size_t f = pStart + 1; f <= pHi; f++) {
            excite_track_setup(ET, C, frames + 16 * (f - 1), frames + 16 * f);
            coef_track_setup(CT, C, frames + 16 * (f - 1), frames + 16 * f);
            for (int j = 0; j < C.controlPeriod; j++) {
                Excitation E = excite_sample(ES, ET, C, C.fir, j, lp[nBase + n], sineLookup);
                Coefs K = coef_sample(CT, C, j);
                sig[25 + n] = tube_sample(TS, C, E, K);
                n++;
            }
        }
        const uint64_t kLo = outputs_through(pLo * CP), kHi = last ? total : outputs_through(pHi * CP);
        for (uint64_t k = kLo; k < kHi; k++) {
            const uint32_t ph = src_phase((uint32_t)k, inc);
            const uint64_t e = ((uint64_t)k * inc) >> 16;          // global read position; local = e - nBase
            float y = src_dot(&sig[e - nBase], &rows[(size_t)ph * kSrcRowC]);
            if (k < cap) out[k] = y;
            const float a = fabsf(y);
            if (a > mx) mx = a;
        }
    }
    *nout = (uint32_t)total;
    *maxv = mx;
    return TRM_OK;
}

// ---------------------------------------------------------------------------------------------
// The small-batch formulation (gnuspeech_amd/csrc/trm_quad.h): four time slots per voice in the
// feed-forward stages (closed-form tracks, oscillator phase as a prefix sum, direct-form FIR) and the
// tube split over four parts.  Same interface as above.
#include "../../gnuspeech_amd/csrc/trm_quad.h"

extern "C" int trm_emul_synthesize_quad(const trm_input_params *p, const float *frames, size_t nframes,
                                        float *out, size_t cap, uint32_t *nout, float *maxv, float *tube)
{
    Const C;
    trm_derived d;
    int rc = build_const(*p, C, d);
    if (rc) return rc;
    if (!C.upsample || C.controlPeriod < kSlots) return TRM_ERANGE;
    static std::vector<float> rows;
    if (rows.empty()) build_src_rows(rows);
    *nout = 0; *maxv = 0.f;
    if (nframes == 0) return TRM_OK;
    const size_t CP = (size_t)C.controlPeriod;
    size_t ntube = (nframes - 1) * CP;
    std::vector<float> lp(ntube + 8);
    {
        double seed = 0.7892347, x1 = 0.0;
        for (size_t i = 0; i < ntube + 8; i++) {
            double prod = seed * 377.0;
            seed = prod - (int)prod;
            double nz = seed - 0.5;
            lp[i] = (float)(nz + x1);
            x1 = nz;
        }
    }
    auto sineLookup = [&](int i) { return sine_table(i); };
    auto frame = [&](size_t i) { return frames + 16 * (i < nframes ? i : nframes - 1); };
    std::vector<float> sig(25 + ntube + 2 * C.padSize + 8, 0.0f);
    // osc history ring, unrolled: (a, b) per tube sample with 26 zero pairs in front
    std::vector<float> ab(2 * (26 + ntube + 8), 0.0f);
    float ca[2][kFirWin], cb[2][kFirWin];
    for (int o = 0; o < 2; o++)
        for (int i = 0; i < kFirWin; i++) { ca[o][i] = fir_window_tap_a(C.fir, o, i); cb[o][i] = fir_window_tap_b(C.fir, o, i); }

    OscSlotTrack OT[kSlots];
    CoefTrack CT[kSlots];
    size_t per[kSlots];              // control period of each slot's current sample
    int jj[kSlots];
    for (int s = 0; s < kSlots; s++) {
        per[s] = 0; jj[s] = s;
        osc_slot_setup(OT[s], C, frame(0), frame(1), s);
        coef_track_setup(CT[s], C, frame(0), frame(1));
    }
    double P = 0.0;
    QuadState<Q4> QS;
    quad_reset(QS);
    for (size_t n0 = 0; n0 < ntube; n0 += kSlots) {
        // ---- osc: slots in parallel, phase by prefix sum
        double two[kSlots], incs[kSlots], axd[kSlots];
        float ax[kSlots], ah1[kSlots];
        for (int s = 0; s < kSlots; s++) {
            if (jj[s] >= (int)CP) {
                jj[s] -= (int)CP; per[s]++;
                osc_slot_setup(OT[s], C, frame(per[s]), frame(per[s] + 1), jj[s]);
                coef_track_setup(CT[s], C, frame(per[s]), frame(per[s] + 1));
            }
            double db = __builtin_fma((double)jj[s], OT[s].glotDelta, OT[s].glot0);
            double a = db >= 60.0 ? 1.0 : OT[s].axGeo;
            axd[s] = db <= 0.0 ? 0.0 : a;
            ax[s] = (float)axd[s];
            ah1[s] = amplitude_f(fma_f((float)jj[s], OT[s].aspDelta, OT[s].aspBase));
            incs[s] = osc_increment(OT[s].f0, C);
            two[s] = incs[s] + incs[s];
        }
        double pre[kSlots];
        for (int s = 0; s < kSlots; s++) pre[s] = two[s];
        for (int s = kSlots - 1; s >= 1; s--) pre[s] += pre[s - 1];                  // x += shift1(x)
        for (int s = kSlots - 1; s >= 2; s--) pre[s] += pre[s - 2];                  // x += shift2(x)
        for (int s = 0; s < kSlots; s++) {
            double pos2 = osc_wrap(P + pre[s]), pos1 = osc_wrap((P + pre[s]) - incs[s]);
            float wa, wb;
            osc_read(C, axd[s], pos1, pos2, sineLookup, wa, wb);
            ab[2 * (26 + n0 + s)] = wa; ab[2 * (26 + n0 + s) + 1] = wb;
            OT[s].f0 *= OT[s].f0Step;
            OT[s].axGeo *= OT[s].axStep;
        }
        P = osc_wrap(P + pre[kSlots - 1]);
        // ---- mix + coef per slot, then the tube serially over the block
        for (int s = 0; s < kSlots && n0 + s < ntube; s++) {
            size_t m = n0 + s;
            int o = (int)(m & 1);
            const float *win = &ab[2 * (26 + m - 24 - o)];
            float pulse = fir_direct(win, ca[o], cb[o]);
            Excitation E = mix_tail(C, ax[s], ah1[s], pulse, lp[m]);
            Coefs K = coef_sample(CT[s], C, jj[s]);
            PartRecord R[4];
            pack_part_records(K, C, R);
            Q4 kk[4], tp[4];
            for (int r = 0; r < 4; r++) {
                kk[r] = Q4(R[0].kk[r], R[1].kk[r], R[2].kk[r], R[3].kk[r]);
                tp[r] = Q4(R[0].tp[r], R[1].tp[r], R[2].tp[r], R[3].tp[r]);
            }
            SharedRecord H;
            pack_shared_record(K, C, H);
            Q4 y = tube_quad_step(QS, C, Q4(E.gin), Q4(E.sig), Q4(E.thr), Q4(H.bpA2), Q4(H.bpB2), Q4(H.bpG2),
                                  Q4P{Q4(H.endK[0]), Q4(H.endK[1])}, Q4P{Q4(H.endOnePlus[0]), Q4(H.endOnePlus[1])},
                                  Q4P{kk[0], kk[1]}, Q4P{kk[2], kk[3]}, Q4P{tp[0], tp[1]}, Q4P{tp[2], tp[3]});
            if (tube) tube[m] = y.v[2];
            sig[25 + m] = y.v[2];
        }
        for (int s = 0; s < kSlots; s++) jj[s] += kSlots;
    }
    uint64_t total = count_outputs(d, ntube);
    float mx = 0.f;
    for (uint64_t k = 0; k < total; k++) {
        uint32_t ph = src_phase((uint32_t)k, C.timeRegisterIncrement);
        uint32_t e = src_position((uint32_t)k, C.timeRegisterIncrement);
        float y = src_dot(&sig[e], &rows[(size_t)ph * kSrcRowC]);
        if (k < cap) out[k] = y;
        float a = fabsf(y);
        if (a > mx) mx = a;
    }
    *nout = (uint32_t)total;
    *maxv = mx;
    return TRM_OK;
}

// the product's FIR taps (trm_setup.cc: kFirHalf) as the kernels get them: c[0..24]
extern "C" void trm_emul_fir_half(double *out)
{
    for (int i = 0; i < kFirUnique; i++) out[i] = kFirHalf[i];
}

// tube_quad_step against tube_step on random state/coefficients: returns the number of mismatching
// values over `iters` steps (0 = the four-part data movement is exact).
extern "C" int trm_emul_quad_selfcheck(const trm_input_params *p, int iters, unsigned seed)
{
    Const C;
    trm_derived d;
    if (build_const(*p, C, d)) return -1;
    srand(seed);
    auto rnd = [&]() { return (float)rand() / (float)RAND_MAX * 2.0f - 1.0f; };
    TubeState TS; tube_reset(TS);
    QuadState<Q4> QS; quad_reset(QS);
    int bad = 0;
    for (int it = 0; it < iters; it++) {
        if (it % 200 == 0) { tube_reset(TS); quad_reset(QS); }   // random coefficients are not a passive tube: keep it finite
        Coefs K;
        for (int i = 0; i < 7; i++) K.td[i] = (1.0f + rnd() * 0.9f) * C.damping;
        K.onePlusK8 = 1.0f + rnd() * 0.9f;
        K.k8a = (K.onePlusK8 - 1.0f) * C.mA10;
        K.alphaU = rnd() * 0.5f + 0.5f;
        K.alphaLR = fma_f(-0.5f, K.alphaU, 1.0f);
        K.ntd1 = (1.0f + rnd() * 0.9f) * C.damping;
        for (int i = 0; i < 8; i++) K.tap[i] = rnd() * 0.1f;
        K.bpBeta = 0.2f + rnd() * 0.2f; K.bpGamma = rnd() * 0.3f; K.bpAlpha = (0.5f - K.bpBeta) * 0.5f;
        Excitation E; E.gin = rnd(); E.sig = rnd(); E.thr = rnd();
        float y0 = tube_sample(TS, C, E, K);
        PartRecord R[4];
        pack_part_records(K, C, R);
        Q4 kk[4], tp[4];
        for (int r = 0; r < 4; r++) {
            kk[r] = Q4(R[0].kk[r], R[1].kk[r], R[2].kk[r], R[3].kk[r]);
            tp[r] = Q4(R[0].tp[r], R[1].tp[r], R[2].tp[r], R[3].tp[r]);
        }
        SharedRecord H;
        pack_shared_record(K, C, H);
        Q4 y = tube_quad_step(QS, C, Q4(E.gin), Q4(E.sig), Q4(E.thr), Q4(H.bpA2), Q4(H.bpB2), Q4(H.bpG2),
                              Q4P{Q4(H.endK[0]), Q4(H.endK[1])}, Q4P{Q4(H.endOnePlus[0]), Q4(H.endOnePlus[1])},
                              Q4P{kk[0], kk[1]}, Q4P{kk[2], kk[3]}, Q4P{tp[0], tp[1]}, Q4P{tp[2], tp[3]});
        if (y.v[2] != y0) { if (bad < 5) fprintf(stderr, "it %d: y %g vs %g\n", it, y.v[2], y0); bad++; }
        // state correspondence
        const Waves &w = TS.w;
        const float exp_[] = {w.oT[0], w.oT[1], w.oB[0], w.oT[2], w.oB[1], w.oT[3], w.oB[2], w.oB[3], w.oT[4], w.nT[0],
                              w.oT[5], w.oB[4], w.oT[6], w.oB[5], w.oT[7], w.oB[6], w.oT[8], w.oB[7], w.oT[9], w.oB[8],
                              w.oB[9], w.nT[1], w.nB[0], w.nT[2], w.nB[1], w.nT[3], w.nB[2], w.nT[4], w.nB[3], w.nT[5],
                              w.nB[4], w.nB[5]};
        const Q4 T0 = QS.TA.x, T2 = QS.TA.y, T1 = QS.TB.x, T3 = QS.TB.y, B0 = QS.BA.x, B2 = QS.BA.y, B1 = QS.BB.x, B3 = QS.BB.y;
        const float got[] = {QS.A0.v[0], T0.v[0], B0.v[0], T1.v[0], B1.v[0], T2.v[0], B2.v[0],
                             QS.jB.v[1], QS.jT.v[1], QS.jN.v[1], T0.v[1], B0.v[1], T1.v[1], B1.v[1],
                             T2.v[1], B2.v[1], T0.v[2], B0.v[2], T1.v[2], B1.v[2], QS.eB.x.v[2],
                             T0.v[3], B0.v[3], T1.v[3], B1.v[3], T2.v[3], B2.v[3], T3.v[3],
                             B3.v[3], T3.v[2], B3.v[2], QS.eB.y.v[2]};
        for (size_t i = 0; i < sizeof(exp_) / sizeof(exp_[0]); i++)
            if (exp_[i] != got[i]) { if (bad < 5) fprintf(stderr, "it %d: state %zu: %g vs %g\n", it, i, got[i], exp_[i]); bad++; }
    }
    return bad;
}

// tube_oct_core (eight parts per voice, trm_oct.h) against tube_step on random state / coefficients: the number of
// mismatching values over `iters` steps (0 = the data movement is exact; a value taken from outside the voice's eight
// lanes would be NaN here).
extern "C" int trm_emul_oct_selfcheck(const trm_input_params *p, int iters, unsigned seed)
{
    Const C;
    trm_derived d;
    if (build_const(*p, C, d)) return -1;
    srand(seed);
    auto rnd = [&]() { return (float)rand() / (float)RAND_MAX * 2.0f - 1.0f; };
    TubeState TS; tube_reset(TS);
    OctState<O8> OS; oct_reset(OS);
    OctLane<O8> L;
    for (int i = 0; i < 8; i++) {
        L.p0.v[i] = i == 0; L.p1.v[i] = i == 1; L.p5.v[i] = i == 5; L.end.v[i] = i == 4 || i == 7;
        L.cf.v[i] = i == 4 ? C.mCoeff : i == 7 ? C.nCoeff : 0.0f;
    }
    float bx1 = 0, bx2 = 0, by1 = 0, by2 = 0, thY = 0;
    int bad = 0;
    for (int it = 0; it < iters; it++) {
        if (it % 200 == 0) { tube_reset(TS); oct_reset(OS); bx1 = bx2 = by1 = by2 = thY = 0; }
        Coefs K;
        for (int i = 0; i < 7; i++) K.td[i] = (1.0f + rnd() * 0.9f) * C.damping;
        K.onePlusK8 = 1.0f + rnd() * 0.9f;
        K.k8a = (K.onePlusK8 - 1.0f) * C.mA10;
        K.alphaU = rnd() * 0.5f + 0.5f;
        K.alphaLR = fma_f(-0.5f, K.alphaU, 1.0f);
        K.ntd1 = (1.0f + rnd() * 0.9f) * C.damping;
        for (int i = 0; i < 8; i++) K.tap[i] = rnd() * 0.1f;
        K.bpBeta = 0.2f + rnd() * 0.2f; K.bpGamma = rnd() * 0.3f; K.bpAlpha = (0.5f - K.bpBeta) * 0.5f;
        Excitation E; E.gin = rnd(); E.sig = rnd(); E.thr = rnd();
        float y0 = tube_sample(TS, C, E, K);
        // the two feed-forward recurrences as the kernel's other waves run them
        SharedRecord H;
        pack_shared_bp(K, H);
        const float fr = bandpass_eval<float>(H.bpA2, H.bpB2, H.bpG2, E.sig, bx2, by1, by2);
        bx2 = bx1; bx1 = E.sig; by2 = by1; by1 = fr;
        const float ty = throat_eval<float>(C, E.thr, thY);
        thY = ty;
        float kk[8][2], tp[5][2];
        pack_oct_k(K, C, kk);
        pack_oct_tap(K, tp);
        O8P k, in;
        for (int q = 0; q < 8; q++) {
            k.x.v[q] = kk[q][0]; k.y.v[q] = kk[q][1];
            in.x.v[q] = q < 5 ? tp[q][0] * fr : 0.0f;
            in.y.v[q] = q < 4 ? tp[q][1] * fr : q == 4 ? K.onePlusK8 : q == 7 ? C.onePlusNK6 : 0.0f;
        }
        O8 y = tube_oct_core(OS, O8(C.damping), O8(C.throatGain), L, O8(E.gin), O8(ty), O8(K.alphaLR), k, in);
        if (!(y.v[4] == y0)) { if (bad < 5) fprintf(stderr, "it %d: y %g vs %g\n", it, y.v[4], y0); bad++; }
        const Waves &w = TS.w;
        const float expT[8][2] = {{w.oT[1], w.oT[2]}, {w.oT[3], w.oT[4]}, {w.oT[5], w.oT[6]}, {w.oT[7], w.oT[8]},
                                  {w.oT[9], 0}, {w.nT[1], w.nT[2]}, {w.nT[3], w.nT[4]}, {w.nT[5], 0}};
        const float expB[8][2] = {{w.oB[0], w.oB[1]}, {w.oB[2], w.oB[3]}, {w.oB[4], w.oB[5]}, {w.oB[6], w.oB[7]},
                                  {w.oB[8], w.oB[9]}, {w.nB[0], w.nB[1]}, {w.nB[2], w.nB[3]}, {w.nB[4], w.nB[5]}};
        for (int q = 0; q < 8; q++) {
            const bool endPart = q == 4 || q == 7;
            if (!(OS.T.x.v[q] == expT[q][0]) || (!endPart && !(OS.T.y.v[q] == expT[q][1])) || !(OS.B.x.v[q] == expB[q][0]) ||
                !(OS.B.y.v[q] == expB[q][1])) {
                if (bad < 5) fprintf(stderr, "it %d: part %d: T {%g %g} vs {%g %g}, B {%g %g} vs {%g %g}\n", it, q, OS.T.x.v[q], OS.T.y.v[q],
                                     expT[q][0], expT[q][1], OS.B.x.v[q], OS.B.y.v[q], expB[q][0], expB[q][1]);
                bad++;
            }
        }
        if (!(OS.A0.v[0] == w.oT[0]) || !(OS.jN.v[1] == w.nT[0])) { if (bad < 5) fprintf(stderr, "it %d: glottis / nasal branch\n", it); bad++; }
    }
    return bad;
}

