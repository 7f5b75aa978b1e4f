//  shim/TRMTubeModel.m -- drop-in replacement for Frameworks/Tube/TRMTubeModel.m.
//
//  Implements the reference's own @interface (Frameworks/Tube/TRMTubeModel.h:29-40, imported from the
//  GnuSpeech tree, not duplicated here) by forwarding to libtrm_hip.so through include/trm_c_api.h.
//  With this file in the Tube target, TRMWavetable / TRMFIRFilter / TRMFilters / TRMSampleRateConverter /
//  TRMRingBuffer / TRMUtility drop out of the build; TRMDataList, TRMInputParameters and TRMParameters (pure
//  data classes) and every caller (TRMSynthesizer.m:118-136, Frameworks/Tube/main.m:30-59, Monet's
//  MSynthesisController.m:456-465, GnuTTSServer's PhoneToSpeech.m:82-87) stay untouched.
//
//  SOURCE ONLY: this container has no Objective-C front end or Foundation headers (SURVEY.md 8c), so the
//  file is reviewed, not compiled, here; the C ABI it calls is what the tests exercise.

#import "TRMTubeModel.h"
#import "TRMDataList.h"
#import "TRMInputParameters.h"
#import "TRMParameters.h"

#include <stdlib.h>
#include "trm_c_api.h"

@implementation TRMTubeModel
{
    TRMDataList *_inputData;
    trm_tube *_tube;
}

static void TRMFillInputParams(trm_input_params *p, TRMInputParameters *ip)
{
    p->outputFileFormat = (int32_t)ip.outputFileFormat;
    p->outputRate       = ip.outputRate;
    p->controlRate      = ip.controlRate;
    p->volume           = ip.volume;
    p->channels         = (int32_t)ip.channels;
    p->balance          = ip.balance;
    p->waveform         = (int32_t)ip.waveform;
    p->tp               = ip.tp;
    p->tnMin            = ip.tnMin;
    p->tnMax            = ip.tnMax;
    p->breathiness      = ip.breathiness;
    p->length           = ip.length;
    p->temperature      = ip.temperature;
    p->lossFactor       = ip.lossFactor;
    p->apScale          = ip.apScale;
    p->mouthCoef        = ip.mouthCoef;
    p->noseCoef         = ip.noseCoef;
    for (int i = 0; i < TRM_TOTAL_NASAL_SECTIONS; i++)
        p->noseRadius[i] = ip.noseRadius[i];
    p->throatCutoff     = ip.throatCutoff;
    p->throatVol        = ip.throatVol;
    p->usesModulation   = ip.usesModulation ? 1 : 0;
    p->mixOffset        = ip.mixOffset;
}

- (id)initWithInputData:(TRMDataList *)inputData;
{
    if ((self = [super init])) {
        _inputData = inputData;
        trm_input_params p;
        TRMFillInputParams(&p, inputData.inputParameters);
        // nil on an illegal tube length, like the reference (the library prints the same message)
        if (trm_tube_create(&p, -1, &_tube) != TRM_OK) {
            if (trm_last_error()[0]) NSLog(@"TRMTubeModel: %s", trm_last_error());
            return nil;
        }
    }
    return self;
}

- (void)dealloc;
{
    trm_tube_destroy(_tube);
}

- (void)synthesize;
{
    NSArray *values = _inputData.values;
    NSUInteger count = [values count];
    trm_parameters *frames = (trm_parameters *)calloc(count ? count : 1, sizeof(trm_parameters));
    NSUInteger index = 0;
    for (TRMParameters *q in values) {
        trm_parameters *f = &frames[index++];
        f->glottalPitch             = q.glottalPitch;
        f->glottalVolume            = q.glottalVolume;
        f->aspirationVolume         = q.aspirationVolume;
        f->fricationVolume          = q.fricationVolume;
        f->fricationPosition        = q.fricationPosition;
        f->fricationCenterFrequency = q.fricationCenterFrequency;
        f->fricationBandwidth       = q.fricationBandwidth;
        for (int i = 0; i < TRM_TOTAL_REGIONS; i++)
            f->radius[i]            = q.radius[i];
        f->velum                    = q.velum;
    }
    if (trm_tube_synthesize(_tube, frames, count) != TRM_OK)
        NSLog(@"TRMTubeModel -synthesize: %s", trm_last_error());
    free(frames);
}

- (BOOL)saveOutputToFile:(NSString *)filename error:(NSError **)error;
{
    return trm_tube_save_output_to_file(_tube, [filename fileSystemRepresentation]) == TRM_OK;
}

- (NSData *)generateWAVData;
{
    size_t length = 0;
    NSParameterAssert(trm_tube_maximum_sample_value(_tube) != 0);      // as the reference (TRMTubeModel.m:511)
    if (trm_tube_generate_wav_data(_tube, NULL, 0, &length) != TRM_OK)
        return nil;
    NSMutableData *data = [NSMutableData dataWithLength:length];
    if (trm_tube_generate_wav_data(_tube, (uint8_t *)[data mutableBytes], length, &length) != TRM_OK)
        return nil;
    return [data copy];
}

- (void)printInputData;
{
    [_inputData printInputParameters];
    trm_derived d;
    if (trm_tube_derived(_tube, &d) == TRM_OK) {
        printf("\nactual tube length:\t%.4f cm\n", d.actualTubeLength);
        printf("internal sample rate:\t%-d Hz\n", d.sampleRate);
        printf("control period:\t\t%-d samples (%.4f seconds)\n\n", d.controlPeriod, (float)d.controlPeriod / (float)d.sampleRate);
    }
    [_inputData printControlRateInputTable];
}

@end
