"""Host-side mirror of the reference's control-track generator, over the C ABI.

    Event        Frameworks/GnuSpeech/MonetModel/Event.h:12-22 (time in ms + 36 values, NaN = no target)
    MMIntonation Frameworks/GnuSpeech/MonetModel/MMIntonation.h:10-16 (the switches this step reads)
    EventList    -generateOutputInTimeRange:forSynthesizer:parameterLogger: (EventList.m:883-1061)

Building the event list (postures, rules, rhythm, the intonation contour) is Monet's rule engine and out
of scope; this module starts where the events exist and turns them into 250 Hz frames ON THE GPU
(libtrm_hip.so: trm_tracks_kernel).  There is no CPU fallback.
"""
import ctypes as C

import numpy as np

from ._capi import TrmIntonation, check, lib

MAX_VALUES = 36


class Event:
    def __init__(self, time):
        self.time = int(time)
        self.values = np.full(MAX_VALUES, np.nan, dtype=np.float64)       # Event.m: NaN-initialised

    def setValue(self, value, index):
        self.values[index] = value

    def getValueAtIndex(self, index):
        return float(self.values[index])


class MMIntonation:
    def __init__(self):
        self.shouldUseMacroIntonation = True
        self.shouldUseMicroIntonation = True
        self.shouldUseSmoothIntonation = True
        self.shouldUseDrift = True
        self.driftDeviation = 1.0
        self.driftCutoff = 4.0


def intonation_struct(intonation, pitch_mean, time_quantization=4, start_ms=0, length_ms=0, drift_seed=0.0):
    s = TrmIntonation()
    s.useMicroIntonation = int(bool(intonation.shouldUseMicroIntonation))
    s.useMacroIntonation = int(bool(intonation.shouldUseMacroIntonation))
    s.useSmoothIntonation = int(bool(intonation.shouldUseSmoothIntonation))
    s.useDrift = int(bool(intonation.shouldUseDrift))
    s.driftDeviation = intonation.driftDeviation
    s.driftCutoff = intonation.driftCutoff
    s.pitchMean = float(pitch_mean)
    s.timeQuantization = int(time_quantization)
    s.startTime_ms = int(start_ms)
    s.endTime_ms = int(start_ms + length_ms) if length_ms else 0        # NSMaxRange; length 0 = everything (:892-894)
    s.driftSeed = float(drift_seed)
    return s


class EventList:
    def __init__(self, pitch_mean=0.0, time_quantization=4):
        self.events = []
        self.intonation = MMIntonation()
        self.pitchMean = pitch_mean                    # model.synthesisParameters.pitch (EventList.m:983)
        self.timeQuantization = time_quantization
        self.driftSeed = 0.0                           # the list's drift generator: 0 = not used yet (MMDriftGenerator.m:27-35)

    def arrays(self):
        times = np.array([e.time for e in self.events], dtype=np.uint32)
        values = (np.stack([e.values for e in self.events]) if self.events else np.zeros((0, MAX_VALUES))).astype(np.float64)
        return times, np.ascontiguousarray(values)

    def settings(self, start_ms=0, length_ms=0):
        return intonation_struct(self.intonation, self.pitchMean, self.timeQuantization, start_ms, length_ms, self.driftSeed)

    def count_frames(self, start_ms=0, length_ms=0):
        times, _ = self.arrays()
        n = C.c_size_t()
        s = self.settings(start_ms, length_ms)
        check(lib().trm_events_count_frames(times.ctypes.data, len(times), C.byref(s), C.byref(n)))
        return n.value

    def generateOutputInTimeRange(self, batch, synthesizer=None, start_ms=0, length_ms=0):
        """Returns the [n,16] float32 frames (computed on the GPU through `batch`, a TRMBatch) and, like the
        reference, hands each of them to `synthesizer.addParameters` when one is given."""
        from .tube import TRMParameters
        times, values = self.arrays()
        s = self.settings(start_ms, length_ms)
        cap = self.count_frames(start_ms, length_ms)
        out = np.zeros((max(cap, 1), 16), dtype=np.float32)
        n = C.c_size_t()
        check(lib().trm_batch_generate_frames_host(batch._h, times.ctypes.data, values.ctypes.data, len(times), C.byref(s),
                                                   out.ctypes.data, cap, C.byref(n)))
        frames = out[:n.value]
        if self.intonation.shouldUseDrift:
            # the generator belongs to the list and keeps its seed from one utterance to the next (EventList.m:105-106,
            # 903; MMDriftGenerator.m:41-58): one -generateDrift per 4 ms step, whatever the time range
            self.driftSeed = float(lib().trm_drift_seed_after(self.driftSeed, self.count_frames()))
        if synthesizer is not None:
            for row in frames:
                synthesizer.addParameters(TRMParameters(row))
        return frames
