// trm_io.h -- host-side formats either side of the tube: the .trm / Monet.parameters text
// format (TRMDataList.m:43-247, MMSynthesisParameters.m:278-310, TRMParameters.m:26-43) and
// the sound-file writers (TRMTubeModel.m:365-593).
#pragma once

#include <stddef.h>
#include <stdint.h>

#include <vector>

#include "../../include/trm_c_api.h"

namespace trm {

double io_amplitude(double decibelLevel);   // TRMUtility.m:26-41

int io_read_data_list(const char *path, trm_input_params &p, std::vector<trm_parameters> &frames);
int io_write_data_list(const char *path, const trm_input_params &p, const trm_parameters *frames, size_t n);

// int16 scaling shared by the writers (TRMTubeModel.m:370-389 / :515-533); out is interleaved
// for 2 channels, host byte order.
void io_scale_int16(const trm_input_params &p, const float *samples, size_t n, double maxSample,
                    bool forWavData, int16_t *out);

// the container's header for n samples into hdr[>= 56]; returns its length (0: unknown format)
size_t io_sound_file_header(const trm_input_params &p, size_t n, uint8_t *hdr);
// -saveOutputToFile:error: (TRMTubeModel.m:365-490): AU / AIFF big-endian, WAVE little-endian.
int io_write_sound_file(const char *path, const trm_input_params &p, const float *samples, size_t n, double maxSample);

// -generateWAVData (TRMTubeModel.m:509-593)
size_t io_wav_data_size(const trm_input_params &p, size_t n);
void io_wav_data(const trm_input_params &p, const float *samples, size_t n, double maxSample, uint8_t *buf);

}  // namespace trm
