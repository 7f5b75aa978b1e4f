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
    ExciteState ES; ExciteTrack ET; CoefTrack CT; TubeState TS;
    excite_reset(ES); tube_reset(TS);
    auto sineLookup = [&](int i) { return sine_table(i); };
    // tube-rate signal, with the converter's 25 zeros of pre-roll and 2*pad zeros of flush around it
    std::vector<float> sig(25 + ntube + 2 * C.padSize, 0.0f);
    size_t n = 0;
    for (size_t f = 1; f < nframes; f++) {
        excite_track_setup(ET, C, frames + 16 * (f - 1), frames + 16 * f);
        coef_track_setup(CT, C, frames + 16 * (f - 1), frames + 16 * f);
        for (int j = 0; j < C.controlPeriod; j++) {
            Excitation E = excite_sample(ES, ET, C, C.fir, j, lp[n], sineLookup);
            Coefs K = coef_sample(CT, C, j);
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
        if (k < cap) out[k] = y;
        float a = fabsf(y);
        if (a > mx) mx = a;
    }
    *nout = (uint32_t)total;
    *maxv = mx;
    return TRM_OK;
}
