// trm_setup.h -- host-side derivation of everything -[TRMTubeModel initWithInputData:] computes
// once per tube (Frameworks/Tube/TRMTubeModel.m:186-260) into the wave-uniform trm::Const,
// plus the sample-rate-converter coefficient tables and the exact output-count bookkeeping.
#pragma once

#include <stddef.h>
#include <stdint.h>

#include <vector>

#include "../../include/trm_c_api.h"
#include "trm_lane.h"

namespace trm {

// TRM_OK or TRM_EINVAL_LENGTH / TRM_EFIR / TRM_ERANGE.
int build_const(const trm_input_params &p, Const &c, trm_derived &d);

extern const double kFirHalf[kFirUnique];      // the 25 distinct taps of the oscillator FIR (trm_setup.cc)

// h[] / deltaH[] of the converter (TRMSampleRateConverter.m:110-131), 3328 doubles each.
void build_src_h(std::vector<double> &h, std::vector<double> &dh);

// Up-sampling coefficient rows, one 128-byte row per 16-bit phase f = (l<<8)|m: 26 coefficients in
// window order (left wing h[l+256i] + deltaH[l+256i]*m/256 reversed, then the right wing at phase ~f;
// TRMSampleRateConverter.m:182-203); 65536 x kSrcRowC floats.
void build_src_rows(std::vector<float> &rows);

// Fine table for the down-sampling branch: fine[q] = h[q>>8] + deltaH[q>>8]*(q&255)/256,
// q < 3328*256 (TRMSampleRateConverter.m:246-270).
constexpr uint32_t kSrcFineLen = 13u * 256u * 256u;    // entries of the fine table: FILTER_LENGTH x 256
void build_src_fine(std::vector<float> &fine);
// Down-sampling branch, per 16-bit phase f of the time register: the coefficients the reference's two wing loops
// (TRMSampleRateConverter.m:243-270) walk through, laid out per output as [left tap 0 .. lmax-1 | right tap 0 .. rmax-1]
// (taps past a wing's own end: 0), `pitch` floats per row.  Same values as trm_downsample_kernel's own walk.
void build_down_rows(const std::vector<float> &fine, double ratio, uint32_t phaseIncrement, uint32_t &lmax,
                     uint32_t &rmax, uint32_t &pitch, std::vector<float> &rows);

// Exact number of converter outputs for a tube that received `ntube` samples, following the
// ring-buffer bookkeeping literally (TRMRingBuffer.m:47-93, TRMSampleRateConverter.m:155-298),
// integer arithmetic only.
uint64_t count_outputs(const trm_derived &d, uint64_t ntube);

// 512-entry sine table (TRMWavetable.m:98-101), fp32.
void build_sine_table(std::vector<float> &tab);

}  // namespace trm
