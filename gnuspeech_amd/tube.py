"""Host-side mirror of the reference's Tube framework classes, over the C ABI.

Same class and method names as the Objective-C originals so that tests read like calls into
the reference:
    TRMInputParameters   Frameworks/Tube/TRMInputParameters.h:26-54
    TRMParameters        Frameworks/Tube/TRMParameters.h:9-17
    TRMDataList          Frameworks/Tube/TRMDataList.h:12-17, TRMDataList.m:32-40
    TRMTubeModel         Frameworks/Tube/TRMTubeModel.h:29-40
    TRMSynthesizer       Frameworks/GnuSpeech/Tube/TRMSynthesizer.h:9-21
All arithmetic happens in libtrm_hip.so on the GPU; these classes only hold data.
"""
import ctypes as C

import numpy as np

from . import _capi
from ._capi import TrmDerived, TrmInputParams, TrmParameters, check, lib

TRMSoundFileFormat_AU, TRMSoundFileFormat_AIFF, TRMSoundFileFormat_WAVE = 0, 1, 2
TRMWaveFormType_Pulse, TRMWaveFormType_Sine = 0, 1

_PARAM_FIELDS = [n for n, _ in TrmInputParams._fields_]


class TRMInputParameters:
    """The 26 utterance-rate fields.  Attribute access goes straight to the POD struct."""

    def __init__(self, **kw):
        object.__setattr__(self, "c", TrmInputParams())
        for k, v in kw.items():
            setattr(self, k, v)

    def __getattr__(self, name):
        if name == "noseRadius":
            return self.c.noseRadius
        if name in _PARAM_FIELDS:
            return getattr(self.c, name)
        raise AttributeError(name)

    def __setattr__(self, name, value):
        if name == "noseRadius":
            for i, v in enumerate(value):
                self.c.noseRadius[i] = v
        elif name in _PARAM_FIELDS:
            setattr(self.c, name, value)
        else:
            raise AttributeError(name)

    @classmethod
    def from_dict(cls, d):
        return cls(**d)


class TRMParameters:
    """One control-rate frame: 7 scalars + radius[8] + velum."""
    __slots__ = ("glottalPitch", "glottalVolume", "aspirationVolume", "fricationVolume", "fricationPosition",
                 "fricationCenterFrequency", "fricationBandwidth", "radius", "velum")

    def __init__(self, values=None):
        v = [0.0] * 16 if values is None else [float(x) for x in values]
        (self.glottalPitch, self.glottalVolume, self.aspirationVolume, self.fricationVolume,
         self.fricationPosition, self.fricationCenterFrequency, self.fricationBandwidth) = v[:7]
        self.radius = list(v[7:15])
        self.velum = v[15]

    def values(self):
        return [self.glottalPitch, self.glottalVolume, self.aspirationVolume, self.fricationVolume,
                self.fricationPosition, self.fricationCenterFrequency, self.fricationBandwidth] + list(self.radius) + [self.velum]

    @property
    def valuesString(self):                      # TRMParameters.m:26-43
        return " ".join("%.3f" % x for x in self.values())


class TRMDataList:
    """inputParameters + values (list of TRMParameters)."""

    def __init__(self):
        self.inputParameters = TRMInputParameters()
        self.values = []

    @classmethod
    def initWithContentsOfFile(cls, path):
        """TRMDataList.m:32-40,43-247.  Returns None on failure like the reference returns nil."""
        self = cls()
        frames = C.POINTER(TrmParameters)()
        n = C.c_size_t()
        rc = lib().trm_data_list_read_file(str(path).encode(), C.byref(self.inputParameters.c), C.byref(frames), C.byref(n))
        if rc != _capi.TRM_OK:
            return None
        if n.value:
            arr = np.ctypeslib.as_array(C.cast(frames, C.POINTER(C.c_double)), shape=(n.value, 16)).copy()
        else:
            arr = np.zeros((0, 16))
        lib().trm_free(frames)
        self.values = [TRMParameters(row) for row in arr]
        return self

    def writeToFile(self, path):
        fr = self.frame_array()
        buf = (TrmParameters * max(1, len(self.values)))()
        if fr.size:
            C.memmove(buf, fr.ctypes.data, fr.nbytes)
        check(lib().trm_data_list_write_file(str(path).encode(), C.byref(self.inputParameters.c), buf, len(self.values)))

    def frame_array(self):
        if not self.values:
            return np.zeros((0, 16), dtype=np.float64)
        return np.ascontiguousarray([p.values() for p in self.values], dtype=np.float64)


class TRMTubeModel:
    """-initWithInputData: / -synthesize / -saveOutputToFile:error: / -generateWAVData."""

    def __init__(self):
        raise TypeError("use TRMTubeModel.initWithInputData(dataList)")

    @classmethod
    def initWithInputData(cls, inputData, device=-1):
        """Returns None when the reference would return nil (length <= 0, TRMTubeModel.m:204-207)."""
        self = object.__new__(cls)
        self.inputData = inputData
        self._h = C.c_void_p()
        rc = lib().trm_tube_create(C.byref(inputData.inputParameters.c), device, C.byref(self._h))
        if rc in (_capi.TRM_EINVAL_LENGTH, _capi.TRM_EFIR):
            self._h = None
            return None
        check(rc)
        return self

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            try:
                lib().trm_tube_destroy(h)
            except Exception:      # interpreter shutdown: the process is going away anyway
                pass
            self._h = None

    def derived(self):
        d = TrmDerived()
        check(lib().trm_tube_derived(self._h, C.byref(d)))
        return {k: getattr(d, k) for k, _ in TrmDerived._fields_}

    def printInputData(self):                    # TRMTubeModel.m:595-605, to stdout like the reference
        import sys
        sys.stdout.flush()
        fr = self.inputData.frame_array()
        buf = (TrmParameters * max(1, fr.shape[0]))()
        if fr.shape[0]:
            C.memmove(buf, fr.ctypes.data, fr.nbytes)
        check(lib().trm_tube_print_input_data(self._h, buf, fr.shape[0]))

    def synthesize(self):
        fr = self.inputData.frame_array()
        buf = (TrmParameters * max(1, fr.shape[0]))()
        if fr.shape[0]:
            C.memmove(buf, fr.ctypes.data, fr.nbytes)
        check(lib().trm_tube_synthesize(self._h, buf, fr.shape[0]))

    @property
    def numberSamples(self):
        return lib().trm_tube_number_samples(self._h)

    @property
    def maximumSampleValue(self):
        return lib().trm_tube_maximum_sample_value(self._h)

    def samples(self):
        n = self.numberSamples
        p = lib().trm_tube_samples(self._h)
        if n == 0 or not p:
            return np.zeros(0, dtype=np.float32)
        return np.ctypeslib.as_array(p, shape=(n,)).copy()

    def saveOutputToFile(self, filename):
        rc = lib().trm_tube_save_output_to_file(self._h, str(filename).encode())
        return rc == _capi.TRM_OK

    def generateWAVData(self):
        n = C.c_size_t()
        check(lib().trm_tube_generate_wav_data(self._h, None, 0, C.byref(n)))
        buf = (C.c_uint8 * n.value)()
        check(lib().trm_tube_generate_wav_data(self._h, buf, n.value, C.byref(n)))
        return bytes(buf)


class TRMSynthesizer:
    """Frameworks/GnuSpeech/Tube/TRMSynthesizer.m: collects frames, builds one tube per utterance."""

    def __init__(self):
        self._inputData = TRMDataList()
        self._inputData.inputParameters.outputFileFormat = 0          # :30
        self.shouldSaveToSoundFile = False
        self.filename = None
        self.lastWAVData = None

    def setupSynthesisParameters(self, sp):
        """sp: mapping with MMSynthesisParameters' names (TRMSynthesizer.m:38-65)."""
        p = self._inputData.inputParameters
        p.outputRate = sp["sampleRate"]
        p.controlRate = 250                                            # :41
        p.volume = sp["masterVolume"]
        p.channels = sp["outputChannels"] + 1                          # :43
        p.balance = sp["balance"]
        p.waveform = sp["glottalPulseShape"]
        p.tp, p.tnMin, p.tnMax = sp["tp"], sp["tnMin"], sp["tnMax"]
        p.breathiness = sp["breathiness"]
        p.length = sp["vocalTractLength"]
        p.temperature = sp["temperature"]
        p.lossFactor = sp["lossFactor"]
        p.apScale = sp["apertureScaling"]
        p.mouthCoef, p.noseCoef = sp["mouthCoef"], sp["noseCoef"]
        p.noseRadius = [0.0, sp["n1"], sp["n2"], sp["n3"], sp["n4"], sp["n5"]]   # :56-61
        p.throatCutoff, p.throatVol = sp["throatCutoff"], sp["throatVolume"]
        p.usesModulation = int(bool(sp["shouldUseNoiseModulation"]))
        p.mixOffset = sp["mixOffset"]

    def removeAllParameters(self):
        self._inputData.values = []

    def addParameters(self, parameters):
        self._inputData.values.append(parameters)                      # no doubling of the last frame (:103-106)

    @property
    def fileType(self):
        return self._inputData.inputParameters.outputFileFormat

    @fileType.setter
    def fileType(self, v):
        self._inputData.inputParameters.outputFileFormat = v

    def synthesize(self):
        tube = TRMTubeModel.initWithInputData(self._inputData)
        if tube is None:
            print("Warning: Failed to create tube model.")
            return None
        tube.synthesize()
        if self.shouldSaveToSoundFile:
            tube.saveOutputToFile(self.filename)
        else:
            self.lastWAVData = tube.generateWAVData()                  # the reference hands this to AVAudioPlayer (:138-153)
        return tube
