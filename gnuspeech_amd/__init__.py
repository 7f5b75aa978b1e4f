"""gnuspeech_amd -- MI355X-native Tube Resonance Model for GnuSpeech (the -[TRMTubeModel synthesize]
hot path).  The arithmetic lives in libtrm_hip.so (hand-written HIP for gfx950); this package is the
host-side mirror of the reference's Tube framework interface plus a batch front end."""
from ._capi import LIB_PATH, TrmError, lib  # noqa: F401
from .batch import TRMBatch, TRMMultiBatch, shard_voices  # noqa: F401
from .events import Event, EventList, MMIntonation, intonation_struct  # noqa: F401
from .stream import TRMStream  # noqa: F401
from .tube import (TRMDataList, TRMInputParameters, TRMParameters, TRMSynthesizer, TRMTubeModel,  # noqa: F401
                   TRMSoundFileFormat_AIFF, TRMSoundFileFormat_AU, TRMSoundFileFormat_WAVE,
                   TRMWaveFormType_Pulse, TRMWaveFormType_Sine)
