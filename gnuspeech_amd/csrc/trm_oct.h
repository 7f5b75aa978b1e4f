// trm_oct.h -- the tube stage with EIGHT lanes per voice (trm_quad.hip's "oct" instance): the same junction
// arithmetic as trm_lane.h's tube_step / trm_quad.h's tube_quad_core (same operations in the same order, so the
// three agree bit for bit), spread over eight "parts" of two junction slots each instead of four parts of four.
//
// Why: a wave issues one instruction per ~4.85 cycles whatever its width (profiles/valu_ceiling_r02.txt), so the tube
// wave's cost per sample is its instruction COUNT.  With four parts a lane carries two packed pairs, the ends as a third
// pair, 10 cross-part moves and 6.5 pair-assembly moves: 50 instructions per sample.  With eight parts a lane carries
// ONE packed pair; the two values a part takes from its neighbours are rotated straight into the pair's halves and the
// special slots (three-way junction, the two ends) are plain fp32 operations on lane-dependent constants.  A wave then
// carries 8 voices, a workgroup of 16 voices has two tube waves, and no SIMD is left with a lone wave.
//
//   part:   0        1        2        3        4        5        6        7
//   slot x  J1       J3       J5       J7       J9       N1       N3       N5
//   slot y  J2       W        J6       J8       M        N2       N4       E
// W = the three-way junction (TRMTubeModel.m:801-806), M / E = the mouth / nose end filters (:820-836, :848-852), J6 the
// junction-less S6|S7 boundary.  A voice's parts are 8 consecutive lanes (two voices per row of 16), so every link of
// the chain is one row rotation: part p takes T.y of part p-1 (a-input of its slot x) and B.x of part p+1 (b-input of
// its slot y); the links that are not neighbours -- the three-way junction's nasal branch, parts 1 <-> 5 -- are
// rotations by 4.  What a rotation brings to a lane that has no such neighbour (part 0's left, part 7's right, the
// other voice of the row) is never used: part 0's a-input is the glottis end, slots M and E have no b-input.
#pragma once

#include "trm_quad.h"

namespace trm {

// ---------------------------------------------------------------- host model of one voice's eight lanes
struct O8 {
    float v[8];
    O8() = default;
    explicit O8(float x) { for (int i = 0; i < 8; i++) v[i] = x; }
};
struct M8 {
    bool v[8];
};
#define TRM_O8_BINOP(op) \
    inline O8 operator op(O8 a, O8 b) { O8 r; for (int i = 0; i < 8; i++) r.v[i] = a.v[i] op b.v[i]; return r; }
TRM_O8_BINOP(+)
TRM_O8_BINOP(-)
TRM_O8_BINOP(*)
#undef TRM_O8_BINOP
inline O8 operator-(O8 a) { O8 r; for (int i = 0; i < 8; i++) r.v[i] = -a.v[i]; return r; }
inline O8 fma_f(O8 a, O8 b, O8 c) { O8 r; for (int i = 0; i < 8; i++) r.v[i] = fma_f(a.v[i], b.v[i], c.v[i]); return r; }
// lane i takes src of lane i - K (mod 16) of its row; a voice is half a row, so what comes from outside the voice is
// the OTHER voice's value: the host model poisons it (a use would show up as NaN in the output)
template <int K>
inline O8 o_rot(O8 src)
{
    O8 r;
    for (int i = 0; i < 8; i++) {
        const int s = (i - K) & 15;
        r.v[i] = s < 8 ? src.v[s] : __builtin_nanf("");
    }
    return r;
}
inline O8 o_sel(M8 m, O8 a, O8 b) { O8 r; for (int i = 0; i < 8; i++) r.v[i] = m.v[i] ? a.v[i] : b.v[i]; return r; }
struct O8P {
    O8 x, y;
};
inline O8P operator+(O8P a, O8P b) { return O8P{a.x + b.x, a.y + b.y}; }
inline O8P operator-(O8P a, O8P b) { return O8P{a.x - b.x, a.y - b.y}; }
inline O8P operator*(O8P a, O8P b) { return O8P{a.x * b.x, a.y * b.y}; }
inline O8P pk_make(O8 x, O8 y) { return O8P{x, y}; }
inline O8P pk_fma(O8P a, O8P b, O8P c) { return O8P{fma_f(a.x, b.x, c.x), fma_f(a.y, b.y, c.y)}; }
template <> struct PairOf<O8> { typedef O8P type; };
template <class F> struct MaskOf;
template <> struct MaskOf<O8> { typedef M8 type; };

#if defined(__HIP__)
template <int K>
__device__ __forceinline__ float o_rot(float src)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, src), 0x120 + K /* row_ror:K */, 0xF, 0xF, false));
}
__device__ __forceinline__ float o_sel(bool m, float a, float b) { return m ? a : b; }
template <> struct MaskOf<float> { typedef bool type; };
#endif

// what a lane is, fixed for the kernel's lifetime
template <class F>
struct OctLane {
    typename MaskOf<F>::type p0, p1, p5, end;   // part 0 (glottis end), part 1 (three-way junction), part 5 (its nasal branch), parts 4 | 7 (ends)
    F cf;                                       // the end filters' coefficient: mCoeff in part 4, nCoeff in part 7, 0 elsewhere
};

// ---------------------------------------------------------------- coefficient records: one float4 per (sample, part)
// {k.x, k.y | in.x, in.y}: transmission factors t = (1 + k) d of the part's two slots and their frication injections
// tap x band-pass output.  The special slots carry what they need in the same places: W {alphaU | tap FC3 x f},
// M {C8 a10 | 1 + C8}, E {NC6 a10 | 1 + NC6}.
TRM_HD void pack_oct_k(const Coefs &K, const Const &C, float k[8][2])          // coef_sample_area's fields
{
    const float t[8][2] = {{K.td[0], K.td[1]}, {K.td[2], K.alphaU}, {K.td[3], C.damping}, {K.td[4], K.td[5]},
                           {K.td[6], K.k8a}, {K.ntd1, C.nasalTd[0]}, {C.nasalTd[1], C.nasalTd[2]}, {C.nasalTd[3], C.nasalK6a}};
    for (int p = 0; p < 8; p++) { k[p][0] = t[p][0]; k[p][1] = t[p][1]; }
}
// taps of parts 0..4 (parts 5..7, the nasal tract, have none); part 4's slot y is the mouth end: its `in` is 1 + C8
TRM_HD void pack_oct_tap(const Coefs &K, float tp[5][2])                       // coef_sample_fric's fields
{
    const float t[5][2] = {{0.0f, K.tap[0]}, {K.tap[1], K.tap[2]}, {K.tap[3], K.tap[4]}, {K.tap[5], K.tap[6]}, {K.tap[7], 0.0f}};
    for (int p = 0; p < 5; p++) { tp[p][0] = t[p][0]; tp[p][1] = t[p][1]; }
}

template <class F>
struct OctState {
    typedef typename PairOf<F>::type P;
    P T, B;                 // this part's junction outputs {T of slot x, T of slot y}, {B of slot x, B of slot y}
    F A0;                   // part 0: oT[0], the glottis end
    F jN;                   // part 1: the three-way junction's nT[0]
    F reflY, radX, radY;    // parts 4 / 7: end filter memories
};

template <class F>
TRM_HD void oct_reset(OctState<F> &S)
{
    const F z = F(0.0f);
    S.T = pk_make(z, z); S.B = pk_make(z, z);
    S.A0 = S.jN = S.reflY = S.radX = S.radY = z;
}

// One tube sample: `gin` the glottal input, `ty` the throat output of that sample, `aLR` the three-way junction's
// alpha-left = alpha-right (voice-wide values; part 0 uses gin, part 1 aLR, part 4 ty: the kernel hands each part ITS
// value in one register), k / in this part's record; d = C.damping, tg = C.throatGain (parameters so that the kernel can hand them over in VECTOR registers:
// an instruction with a scalar-register operand does not co-issue with another wave's, profiles/valu_ceiling_r02.txt).
// Returns the tube-rate output in PART 4 (other parts: unspecified).
template <class F>
TRM_HD F tube_oct_core(OctState<F> &S, F d, F tg, const OctLane<F> &L, F gin, F ty, F aLR, typename PairOf<F>::type k,
                       typename PairOf<F>::type in)
{
    typedef typename PairOf<F>::type P;
    // ---- gather the junctions' inputs from the previous sample's outputs
    F aX = o_rot<1>(S.T.y);                         // slot x's a-input: T of the part before
    aX = o_sel(L.p0, S.A0, aX);                     //   J1 <- the glottis end
    aX = o_sel(L.p5, o_rot<4>(S.jN), aX);           //   N1 <- the three-way junction's nT[0] (part 1)
    const F bY = o_rot<15>(S.B.x);                  // slot y's b-input: B of the part after (W: oB[4] = J5.B)
    const F x3 = o_rot<12>(S.B.x);                  // part 1: nB[0] = N1.B (part 5)
    const F tX = S.T.x;                             // slot y's a-input; W: oT[3] = J3.T; M, E: oT[9] = J9.T, nT[5] = N5.T
    const P aP = pk_make(aX, tX), bP = pk_make(S.B.y, bY);
    // ---- generic junctions (:783-816, :838-846): T = d b + t (a - b) + tap fr, B = T - d (a - b)
    const P dd = pk_make(d, d), ndd = pk_make(-d, -d);
    const P df = aP - bP;
    const P TP = pk_fma(k, df, dd * bP);
    const P BP = pk_fma(ndd, df, TP);
    const P TI = TP + in;
    // ---- glottis end (:781)
    S.A0 = S.B.x * d + gin;
    // ---- three-way junction (:801-806); the three alphas sum to 2 (:733-736)
    const F aU = k.y;
    const F jp = fma_f(aLR, tX, fma_f(aLR, bY, aU * x3));
    const F jB = (jp - tX) * d;
    const F jT = (jp - bY) * d + in.y;
    S.jN = (jp - x3) * d;
    // ---- mouth and nose ends: reflection + radiation (:820-836, :848-852, TRMFilters.m:47-60)
    const F refl = fma_f(k.y, tX, L.cf * S.reflY);
    S.reflY = refl;
    const F eB = d * refl;
    const F rin = in.y * tX;
    const F rad = L.cf * (rin - S.radX + S.radY);
    S.radX = rin; S.radY = rad;
    // ---- commit: the special slots replace the generic results of their slot y
    S.T = pk_make(TI.x, o_sel(L.p1, jT, TI.y));
    S.B = pk_make(BP.x, o_sel(L.p1, jB, o_sel(L.end, eB, BP.y)));
    // ---- the output sum (:346): mouth (part 4) + nose (part 7), then the throat
    const F out = rad + o_rot<13>(rad);
    return ty * tg + out;
}

}  // namespace trm
