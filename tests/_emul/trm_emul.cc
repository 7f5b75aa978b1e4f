// trm_emul.cc -- TEST INFRASTRUCTURE.  Runs gnuspeech_amd/csrc/trm_lane.h (the exact per-lane
// stage arithmetic the HIP kernel executes) serially on the host, one voice at a time, so the
// fp32/fp64 precision plan can be checked against the oracle in a container without a GPU.  Never
// linked into libtrm_hip.so and never used by the product path.
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "../../gnuspeech_amd/csrc/trm_lane.h"
#include "../../gnuspeech_amd/csrc/trm_setup.h"

using namespace trm;

extern "C" int trm_emul_synthesize(const trm_input_params *p, const float *frames, size_t nframes,
                                   float *out, size_t cap, uint32_t *nout, float *maxv, float *tube)
{
    Const C;
    trm_derived d;
    int rc = build_const(*p, C, d);
    if (rc) return rc;
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
    ExciteState ES; ExciteTrack ET; CoefTrack CT; TubeState TS; SrcState<1> SS;
    excite_reset(ES); tube_reset(TS); src_reset(SS);
    uint32_t t = 0; uint64_t e = 0, n = 0, k = 0;
    auto sineLookup = [&](int i) { return sine[i]; };
    auto push = [&](float s) {
        src_push_block<1>(SS, &s);
        while (e <= n) {
            uint32_t f = t & 0xFFFF;
            float y = src_emit_up<1, 0>(SS, &rows[(size_t)f * kSrcRow], &rows[(size_t)(0xFFFF - f) * kSrcRow]);
            if (k < cap) out[k] = y;
            k++;
            float a = fabsf(y);
            if (a > SS.maxAbs) SS.maxAbs = a;
            t += C.timeRegisterIncrement;
            e += t >> 16;
            t &= 0xFFFF;
        }
        n++;
    };
    for (size_t f = 1; f < nframes; f++) {
        excite_track_setup(ET, C, frames + 16 * (f - 1), frames + 16 * f);
        coef_track_setup(CT, C, frames + 16 * (f - 1), frames + 16 * f);
        for (int j = 0; j < C.controlPeriod; j++) {
            Excitation E = excite_sample(ES, ET, C, C.fir, j, lp[n], sineLookup);
            Coefs K = coef_sample(CT, C, j);
            float s = tube_sample(TS, C, E, K);
            if (tube) tube[n] = s;
            push(s);
        }
    }
    for (int i = 0; i < 2 * C.padSize; i++) push(0.0f);
    *nout = (uint32_t)k;
    *maxv = SS.maxAbs;
    return TRM_OK;
}
