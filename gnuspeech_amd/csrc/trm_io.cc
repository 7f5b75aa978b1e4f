// trm_io.cc -- see trm_io.h.
#include "trm_io.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

namespace trm {

double io_amplitude(double db)
{
    db -= 60.0;
    if (db <= -60.0) return 0.0;
    if (db >= 0.0) return 1.0;
    return pow(10.0, db / 20.0);
}

// ---------------------------------------------------------------- text format
int io_read_data_list(const char *path, trm_input_params &p, std::vector<trm_parameters> &frames)
{
    frames.clear();
    memset(&p, 0, sizeof p);
    FILE *fp = fopen(path, "r");
    if (!fp) {
        fprintf(stderr, "Can't open input file \"%s\".\n", path);      // TRMDataList.m:47
        return TRM_EIO;
    }
    char line[128];
    // 26 utterance-rate lines: the first token of each line is the value (TRMDataList.m:53-214)
    struct Field { int kind; void *dst; const char *what; };
    const Field fields[] = {
        {0, &p.outputFileFormat, "output file format"}, {2, &p.outputRate, "output sample rate"},
        {2, &p.controlRate, "input control rate"}, {1, &p.volume, "master volume"},
        {0, &p.channels, "number of sound output channels"}, {1, &p.balance, "stereo balance"},
        {0, &p.waveform, "glottal source waveform type"}, {1, &p.tp, "glottal pulse rise time (tp)"},
        {1, &p.tnMin, "glottal pulse fall time minimum (tnMin)"}, {1, &p.tnMax, "glottal pulse fall time maximum (tnMax)"},
        {1, &p.breathiness, "glottal source breathiness"}, {1, &p.length, "nominal tube length"},
        {1, &p.temperature, "tube temperature"}, {1, &p.lossFactor, "junction loss factor"},
        {1, &p.apScale, "aperture scaling radius"}, {1, &p.mouthCoef, "mouth aperture coefficient"},
        {1, &p.noseCoef, "nose aperture coefficient"}, {1, &p.noseRadius[1], "nose radius 1"},
        {1, &p.noseRadius[2], "nose radius 2"}, {1, &p.noseRadius[3], "nose radius 3"},
        {1, &p.noseRadius[4], "nose radius 4"}, {1, &p.noseRadius[5], "nose radius 5"},
        {1, &p.throatCutoff, "throat lowpass filter cutoff"}, {1, &p.throatVol, "throat volume"},
        {3, &p.usesModulation, "pulse modulation of noise flag"}, {1, &p.mixOffset, "noise crossmix offset"},
    };
    for (const Field &f : fields) {
        if (!fgets(line, sizeof line, fp)) {
            fprintf(stderr, "Can't read %s.\n", f.what);
            fclose(fp);
            return TRM_EPARSE;
        }
        switch (f.kind) {
        case 0: *(int32_t *)f.dst = (int32_t)strtol(line, nullptr, 10); break;
        case 1: *(double *)f.dst = strtod(line, nullptr); break;
        case 2: *(float *)f.dst = (float)strtod(line, nullptr); break;
        case 3: *(int32_t *)f.dst = strtol(line, nullptr, 10) != 0; break;
        }
    }
    while (fgets(line, sizeof line, fp)) {                               // :217-236
        trm_parameters fr;
        double *v = reinterpret_cast<double *>(&fr);
        char *ptr = line;
        for (int i = 0; i < TRM_FRAME_VALUES; i++) v[i] = strtod(ptr, &ptr);
        frames.push_back(fr);
    }
    if (!frames.empty()) frames.push_back(frames.back());               // :239-241 last row doubled
    fclose(fp);
    return TRM_OK;
}

int io_write_data_list(const char *path, const trm_input_params &p, const trm_parameters *frames, size_t n)
{
    FILE *fp = fopen(path, "w");
    if (!fp) return TRM_EIO;
    // MMSynthesisParameters.m:282-307 (it always writes format 0 and rate 250; here the struct's values)
    fprintf(fp, "%u\t\t; %s\n", (unsigned)p.outputFileFormat, "output file format (0 = AU, 1 = AIFF, 2 = WAVE)");
    fprintf(fp, "%g\t\t; %s\n", (double)p.outputRate, "output sample rate (22050.0, 44100.0)");
    fprintf(fp, "%g\t\t; %s\n", (double)p.controlRate, "input control rate (1 - 1000 Hz)");
    fprintf(fp, "%f\t; %s\n", p.volume, "master volume (0 - 60 dB)");
    fprintf(fp, "%lu\t\t; %s\n", (unsigned long)p.channels, "number of sound output channels (1 or 2)");
    fprintf(fp, "%f\t; %s\n", p.balance, "stereo balance (-1 to +1)");
    fprintf(fp, "%lu\t\t; %s\n", (unsigned long)p.waveform, "glottal source waveform type (0 = pulse, 1 = sine)");
    fprintf(fp, "%f\t; %s\n", p.tp, "glottal pulse rise time (5 - 50 % of GP period)");
    fprintf(fp, "%f\t; %s\n", p.tnMin, "glottal pulse fall time minimum (5 - 50 % of GP period)");
    fprintf(fp, "%f\t; %s\n", p.tnMax, "glottal pulse fall time maximum (5 - 50 % of GP period)");
    fprintf(fp, "%f\t; %s\n", p.breathiness, "glottal source breathiness (0 - 10 % of GS amplitude)");
    fprintf(fp, "%f\t; %s\n", p.length, "nominal tube length (10 - 20 cm)");
    fprintf(fp, "%f\t; %s\n", p.temperature, "tube temperature (25 - 40 degrees celsius)");
    fprintf(fp, "%f\t; %s\n", p.lossFactor, "junction loss factor (0 - 5 % of unity gain)");
    fprintf(fp, "%f\t; %s\n", p.apScale, "aperture scaling radius (3.05 - 12 cm)");
    fprintf(fp, "%f\t; %s\n", p.mouthCoef, "mouth aperture coefficient (0 - 0.99)");
    fprintf(fp, "%f\t; %s\n", p.noseCoef, "nose aperture coefficient (0 - 0.99)");
    for (int i = 1; i < TRM_TOTAL_NASAL_SECTIONS; i++)
        fprintf(fp, "%f\t; radius of nose section %d (0 - 3 cm)\n", p.noseRadius[i], i);
    fprintf(fp, "%f\t; %s\n", p.throatCutoff, "throat lowpass frequency cutoff (50 - nyquist Hz)");
    fprintf(fp, "%f\t; %s\n", p.throatVol, "throat volume (0 - 48 dB)");
    fprintf(fp, "%d\t\t; %s\n", p.usesModulation ? 1 : 0, "pulse modulation of noise (0 = off, 1 = on)");
    fprintf(fp, "%f\t; %s\n", p.mixOffset, "noise crossmix offset (30 - 60 db)");
    for (size_t r = 0; r < n; r++) {                                     // TRMParameters.m:26-43
        const double *v = reinterpret_cast<const double *>(&frames[r]);
        for (int i = 0; i < TRM_FRAME_VALUES; i++) fprintf(fp, i ? " %.3f" : "%.3f", v[i]);
        fputc('\n', fp);
    }
    if (fclose(fp) != 0) return TRM_EIO;
    return TRM_OK;
}

// ---------------------------------------------------------------- output scaling + containers
// out-of-range values wrap modulo 2^16 like the reference's x86 build (stereo file path over-drives by up to 2x)
static inline int16_t wrap16(double v) { return (int16_t)(uint16_t)(int64_t)v; }

void io_scale_int16(const trm_input_params &p, const float *s, size_t n, double maxSample, bool forWavData, int16_t *out)
{
    double scale = (32767.0 / maxSample) * io_amplitude(p.volume);      // TRMTubeModel.m:370,515
    if (p.channels == 2) {
        double g = forWavData ? 1.0 : 2.0;                               // :382-383 vs :532-533
        double left = -((p.balance / 2.0) - 0.5) * scale * g;
        double right = ((p.balance / 2.0) + 0.5) * scale * g;
        for (size_t i = 0; i < n; i++) {
            out[2 * i] = wrap16(rint((double)s[i] * left));
            out[2 * i + 1] = wrap16(rint((double)s[i] * right));
        }
    } else {
        for (size_t i = 0; i < n; i++) out[i] = wrap16(rint((double)s[i] * scale));
    }
}

namespace {

void be32(uint8_t *b, uint32_t v) { b[0] = v >> 24; b[1] = v >> 16; b[2] = v >> 8; b[3] = v; }
void be16(uint8_t *b, uint16_t v) { b[0] = v >> 8; b[1] = v & 0xff; }
void le32(uint8_t *b, uint32_t v) { b[0] = v; b[1] = v >> 8; b[2] = v >> 16; b[3] = v >> 24; }
void le16(uint8_t *b, uint16_t v) { b[0] = v & 0xff; b[1] = v >> 8; }

// IEEE 754 80-bit extended, big-endian (AIFF COMM sample rate)
void ext80(uint8_t *b, double v)
{
    memset(b, 0, 10);
    if (v <= 0) return;
    int e;
    double m = frexp(v, &e);                     // v = m * 2^e, m in [0.5,1)
    uint16_t expo = (uint16_t)(e - 1 + 16383);
    uint64_t mant = (uint64_t)ldexp(m, 64);      // explicit integer bit set
    be16(b, expo);
    for (int i = 0; i < 8; i++) b[2 + i] = (uint8_t)(mant >> (56 - 8 * i));
}

}  // namespace

// the container's header for `n` samples (TRMTubeModel.m:414-487 through AudioToolbox in the reference; byte layouts of the
// three formats as this library writes them); returns its length, 0 for an unknown format
size_t io_sound_file_header(const trm_input_params &p, size_t n, uint8_t *hdr)
{
    const int ch = p.channels == 2 ? 2 : 1;
    const size_t bytes = n * ch * 2;
    const uint32_t rate = (uint32_t)p.outputRate;
    if (p.outputFileFormat == TRM_SOUND_FILE_FORMAT_AU) {
        be32(&hdr[0], 0x2e736e64);               // ".snd"
        be32(&hdr[4], 24);
        be32(&hdr[8], (uint32_t)bytes);
        be32(&hdr[12], 3);                       // 16-bit linear PCM
        be32(&hdr[16], rate);
        be32(&hdr[20], (uint32_t)ch);
        return 24;
    }
    if (p.outputFileFormat == TRM_SOUND_FILE_FORMAT_AIFF) {
        memcpy(&hdr[0], "FORM", 4);
        be32(&hdr[4], (uint32_t)(4 + 8 + 18 + 8 + 8 + bytes));
        memcpy(&hdr[8], "AIFF", 4);
        memcpy(&hdr[12], "COMM", 4);
        be32(&hdr[16], 18);
        be16(&hdr[20], (uint16_t)ch);
        be32(&hdr[22], (uint32_t)n);
        be16(&hdr[26], 16);
        ext80(&hdr[28], (double)p.outputRate);
        memcpy(&hdr[38], "SSND", 4);
        be32(&hdr[42], (uint32_t)(8 + bytes));
        be32(&hdr[46], 0);
        be32(&hdr[50], 0);
        return 54;
    }
    if (p.outputFileFormat == TRM_SOUND_FILE_FORMAT_WAVE) {
        memcpy(&hdr[0], "RIFF", 4);
        le32(&hdr[4], (uint32_t)(36 + bytes));
        memcpy(&hdr[8], "WAVEfmt ", 8);
        le32(&hdr[16], 16);
        le16(&hdr[20], 1);
        le16(&hdr[22], (uint16_t)ch);
        le32(&hdr[24], rate);
        le32(&hdr[28], rate * 2 * ch);
        le16(&hdr[32], (uint16_t)(2 * ch));
        le16(&hdr[34], 16);
        memcpy(&hdr[36], "data", 4);
        le32(&hdr[40], (uint32_t)bytes);
        return 44;
    }
    return 0;
}

int io_write_sound_file(const char *path, const trm_input_params &p, const float *s, size_t n, double maxSample)
{
    int ch = p.channels == 2 ? 2 : 1;
    std::vector<int16_t> pcm(n * ch + 1);
    io_scale_int16(p, s, n, maxSample, false, pcm.data());
    size_t bytes = n * ch * 2;
    std::vector<uint8_t> body(bytes);
    bool little = p.outputFileFormat == TRM_SOUND_FILE_FORMAT_WAVE;     // :410-412
    for (size_t i = 0; i < n * ch; i++) {
        if (little) le16(&body[2 * i], (uint16_t)pcm[i]);
        else be16(&body[2 * i], (uint16_t)pcm[i]);
    }
    std::vector<uint8_t> hdr(56);
    hdr.resize(io_sound_file_header(p, n, hdr.data()));
    if (hdr.empty()) return TRM_EINVAL;
    FILE *fp = fopen(path, "wb");
    if (!fp) return TRM_EIO;
    bool ok = fwrite(hdr.data(), 1, hdr.size(), fp) == hdr.size() && (bytes == 0 || fwrite(body.data(), 1, bytes, fp) == bytes);
    if (fclose(fp) != 0) ok = false;
    return ok ? TRM_OK : TRM_EIO;
}

size_t io_wav_data_size(const trm_input_params &p, size_t n)
{
    int ch = p.channels == 2 ? 2 : 1;
    return 12 + (8 + 18) + 8 + n * 2 * ch;
}

void io_wav_data(const trm_input_params &p, const float *s, size_t n, double maxSample, uint8_t *b)
{
    int ch = p.channels == 2 ? 2 : 1;
    size_t bytes = n * 2 * ch;
    int frameSize = (int)ceil(p.channels * (16.0 / 8));                  // TRMTubeModel.m:562-563
    int bytesPerSecond = (int)ceil(p.outputRate * frameSize);
    be32(b, 0x52494646); b += 4;                                         // :571-588
    le32(b, (uint32_t)(4 + (8 + 18) + (8 + bytes))); b += 4;
    be32(b, 0x57415645); b += 4;
    be32(b, 0x666d7420); b += 4;
    le32(b, 18); b += 4;
    le16(b, 1); b += 2;
    le16(b, (uint16_t)p.channels); b += 2;
    le32(b, (uint32_t)p.outputRate); b += 4;
    le32(b, (uint32_t)bytesPerSecond); b += 4;
    le16(b, (uint16_t)frameSize); b += 2;
    le16(b, 16); b += 2;
    le16(b, 0); b += 2;
    be32(b, 0x64617461); b += 4;
    le32(b, (uint32_t)bytes); b += 4;
    std::vector<int16_t> pcm(n * ch + 1);
    io_scale_int16(p, s, n, maxSample, true, pcm.data());
    for (size_t i = 0; i < n * ch; i++) { le16(b, (uint16_t)pcm[i]); b += 2; }
}

}  // namespace trm
