"""ctypes binding of oracle/libtrm_oracle.so -- TEST INFRASTRUCTURE ONLY.

The oracle is the CPU restatement of Frameworks/Tube used as the parity checker.  Only
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB_PATH = os.path.join(ORACLE_DIR, "libtrm_oracle.so")
REF_BIN = os.path.join(ORACLE_DIR, "_ref", "tube_ref")


class InputParams(C.Structure):
    """trm_input_params (include/trm_c_api.h) == TRMInputParameters.h:26-54."""
    _fields_ = [
        ("outputFileFormat", C.c_int32), ("outputRate", C.c_float), ("controlRate", C.c_float),
        ("volume", C.c_double), ("channels", C.c_int32), ("balance", C.c_double),
        ("waveform", C.c_int32), ("tp", C.c_double), ("tnMin", C.c_double), ("tnMax", C.c_double),
        ("breathiness", C.c_double), ("length", C.c_double), ("temperature", C.c_double),
        ("lossFactor", C.c_double), ("apScale", C.c_double), ("mouthCoef", C.c_double),
        ("noseCoef", C.c_double), ("noseRadius", C.c_double * 6), ("throatCutoff", C.c_double),
        ("throatVol", C.c_double), ("usesModulation", C.c_int32), ("mixOffset", C.c_double),
    ]

    def as_dict(self):
        d = {}
        for name, _ in self._fields_:
            v = getattr(self, name)
            d[name] = list(v) if name == "noseRadius" else v
        return d

    @classmethod
    def from_dict(cls, d):
        p = cls()
        for name, _ in cls._fields_:
            if name == "noseRadius":
                for i, v in enumerate(d[name]):
                    p.noseRadius[i] = v
            else:
                setattr(p, name, d[name])
        return p


class Derived(C.Structure):
    _fields_ = [
        ("controlPeriod", C.c_int32), ("sampleRate", C.c_int32), ("actualTubeLength", C.c_double),
        ("sampleRateRatio", C.c_double), ("timeRegisterIncrement", C.c_uint32),
        ("phaseIncrement", C.c_uint32), ("padSize", C.c_int32), ("firTaps", C.c_int32),
    ]


class _Result(C.Structure):
    _fields_ = [
        ("samples", C.POINTER(C.c_double)), ("numberSamples", C.c_int32),
        ("maximumSampleValue", C.c_double), ("tubeSamples", C.POINTER(C.c_double)),
        ("numberTubeSamples", C.c_int32), ("derived", Derived),
    ]


_lib = None


def build():
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR, "libtrm_oracle.so"])


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        L = C.CDLL(LIB_PATH)
        L.trm_oracle_synthesize.argtypes = [C.POINTER(InputParams), C.POINTER(C.c_double), C.c_size_t,
                                            C.c_int, C.POINTER(_Result)]
        L.trm_oracle_synthesize.restype = C.c_int
        L.trm_oracle_synthesize_tract.argtypes = L.trm_oracle_synthesize.argtypes
        L.trm_oracle_synthesize_tract.restype = C.c_int
        L.trm_oracle_synthesize_tract_slices.argtypes = L.trm_oracle_synthesize.argtypes[:3] + [C.c_int32] + L.trm_oracle_synthesize.argtypes[3:]
        L.trm_oracle_synthesize_tract_slices.restype = C.c_int
        L.trm_oracle_result_free.argtypes = [C.POINTER(_Result)]
        L.trm_oracle_run_voices.argtypes = [C.POINTER(InputParams), C.POINTER(C.c_double), C.c_size_t, C.c_size_t,
                                            C.c_size_t, C.c_size_t, C.POINTER(C.c_uint64)]
        L.trm_oracle_run_voices.restype = C.c_int
        L.trm_oracle_derive.argtypes = [C.POINTER(InputParams), C.POINTER(Derived)]
        L.trm_oracle_derive.restype = C.c_int
        L.trm_oracle_fir_taps.argtypes = [C.c_double, C.c_double, C.c_double, C.POINTER(C.c_double), C.c_int]
        L.trm_oracle_fir_taps.restype = C.c_int
        L.trm_oracle_src_tables.argtypes = [C.POINTER(C.c_double), C.POINTER(C.c_double)]
        L.trm_oracle_lp_noise.argtypes = [C.POINTER(C.c_double), C.c_size_t]
        L.trm_oracle_amplitude.argtypes = [C.c_double]
        L.trm_oracle_amplitude.restype = C.c_double
        L.trm_oracle_frequency.argtypes = [C.c_double]
        L.trm_oracle_frequency.restype = C.c_double
        L.trm_oracle_scale_int16.argtypes = [C.POINTER(InputParams), C.POINTER(C.c_double), C.c_int32,
                                             C.c_double, C.c_int, C.POINTER(C.c_int16)]
        L.trm_oracle_wav_data.argtypes = [C.POINTER(InputParams), C.POINTER(C.c_double), C.c_int32,
                                          C.c_double, C.POINTER(C.c_uint8), C.c_size_t]
        L.trm_oracle_wav_data.restype = C.c_size_t
        L.trm_oracle_count_frames.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.POINTER(C.c_size_t)]
        L.trm_oracle_generate_frames.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_size_t,
                                                 C.POINTER(C.c_size_t)]
        L.trm_oracle_parse_file.argtypes = [C.c_char_p, C.POINTER(InputParams),
                                            C.POINTER(C.POINTER(C.c_double)), C.POINTER(C.c_size_t)]
        L.trm_oracle_parse_file.restype = C.c_int
        _lib = L
    return _lib


def _dptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def synthesize(params, frames, keep_tube=False, tract=False, slice=0):
    """Run the oracle.  frames: [n,16] float64.  Returns dict(samples, numberSamples,
    maximumSampleValue, tubeSamples, derived).  tract: in the loop order of Applications/TRAcT/tube.c (frame f >= 1 held
    for control period f, x10 frication taps, x100 before the converter; oracle/trm_oracle.h)."""
    frames = np.ascontiguousarray(frames, dtype=np.float64).reshape(-1, 16)
    res = _Result()
    if tract and slice:          # every frame held for `slice` tube samples (trm_oracle.h)
        rc = lib().trm_oracle_synthesize_tract_slices(C.byref(params), _dptr(frames), frames.shape[0], int(slice), int(keep_tube), C.byref(res))
    else:
        fn = lib().trm_oracle_synthesize_tract if tract else lib().trm_oracle_synthesize
        rc = fn(C.byref(params), _dptr(frames), frames.shape[0], int(keep_tube), C.byref(res))
    if rc != 0:
        raise RuntimeError("trm_oracle_synthesize rc=%d" % rc)
    n = res.numberSamples
    out = {
        "samples": np.ctypeslib.as_array(res.samples, shape=(n,)).copy() if n else np.zeros(0),
        "numberSamples": n,
        "maximumSampleValue": res.maximumSampleValue,
        "derived": {k: getattr(res.derived, k) for k, _ in Derived._fields_},
    }
    nt = res.numberTubeSamples
    out["tubeSamples"] = np.ctypeslib.as_array(res.tubeSamples, shape=(nt,)).copy() if nt else np.zeros(0)
    lib().trm_oracle_result_free(C.byref(res))
    return out


def derive(params):
    d = Derived()
    rc = lib().trm_oracle_derive(C.byref(params), C.byref(d))
    return rc, {k: getattr(d, k) for k, _ in Derived._fields_}


def fir_taps(beta=0.2, gamma=0.1, cutoff=0.00000001):
    buf = np.zeros(401)
    n = lib().trm_oracle_fir_taps(beta, gamma, cutoff, _dptr(buf), 401)
    return buf[:n].copy() if n > 0 else n


def src_tables():
    h = np.zeros(3328)
    dh = np.zeros(3328)
    lib().trm_oracle_src_tables(_dptr(h), _dptr(dh))
    return h, dh


def lp_noise(n):
    a = np.zeros(n)
    lib().trm_oracle_lp_noise(_dptr(a), n)
    return a


def scale_int16(params, samples, maxv, for_wav_data=False):
    samples = np.ascontiguousarray(samples, dtype=np.float64)
    ch = 2 if params.channels == 2 else 1
    out = np.zeros(len(samples) * ch, dtype=np.int16)
    lib().trm_oracle_scale_int16(C.byref(params), _dptr(samples), len(samples), maxv, int(for_wav_data),
                                 out.ctypes.data_as(C.POINTER(C.c_int16)))
    return out


def wav_data(params, samples, maxv):
    samples = np.ascontiguousarray(samples, dtype=np.float64)
    cap = 64 + len(samples) * 4
    buf = np.zeros(cap, dtype=np.uint8)
    n = lib().trm_oracle_wav_data(C.byref(params), _dptr(samples), len(samples), maxv,
                                  buf.ctypes.data_as(C.POINTER(C.c_uint8)), cap)
    return bytes(buf[:n])


def parse_file(path):
    p = InputParams()
    fr = C.POINTER(C.c_double)()
    n = C.c_size_t()
    rc = lib().trm_oracle_parse_file(path.encode(), C.byref(p), C.byref(fr), C.byref(n))
    if rc != 0:
        raise RuntimeError("trm_oracle_parse_file rc=%d" % rc)
    frames = np.ctypeslib.as_array(fr, shape=(n.value, 16)).copy() if n.value else np.zeros((0, 16))
    C.CDLL(None).free(fr)
    return p, frames


# ------------------------------------------------------------------ reference binary (this container only)
def have_ref():
    return os.path.exists(REF_BIN)


def run_ref(params, frames, workdir, tract=False, slice=0):
    """Run oracle/_ref/tube_ref (the reference's own tube.c behind oracle/ref_driver.c); tract: in TRAcT's own
    sample-loop order (tube.c:1096-1190: held parameters, x10 taps, x100 gain) instead of Frameworks/Tube's."""
    frames = np.ascontiguousarray(frames, dtype=np.float64).reshape(-1, 16)
    case = os.path.join(workdir, "case.bin")
    outp = os.path.join(workdir, "out.bin")
    with open(case, "wb") as f:
        f.write(bytes(params))
        f.write(np.uint64(frames.shape[0]).tobytes())
        f.write(frames.tobytes())
    subprocess.check_call([REF_BIN, case, outp] + (["tract"] if tract else []) + (["slice=%d" % slice] if tract and slice else []))
    raw = open(outp, "rb").read()
    o = 0

    def take(dt, n=1):
        nonlocal o
        a = np.frombuffer(raw, dtype=dt, count=n, offset=o)
        o += a.nbytes
        return a
    r = {}
    r["controlPeriod"], r["sampleRate"], r["padSize"], r["firTaps"] = (int(x) for x in take(np.int32, 4))
    r["timeRegisterIncrement"], r["phaseIncrement"] = (int(x) for x in take(np.uint32, 2))
    r["numberSamples"] = int(take(np.int64)[0])
    r["maximumSampleValue"] = float(take(np.float64)[0])
    r["tap_err"] = float(take(np.float64)[0])
    ntube = int(take(np.int64)[0])
    nout = int(take(np.int64)[0])
    r["firCoef"] = take(np.float64, r["firTaps"]).copy()
    r["h"] = take(np.float64, 3328).copy()
    r["deltaH"] = take(np.float64, 3328).copy()
    r["tubeSamples"] = take(np.float64, ntube).copy()
    r["samples_f32"] = take(np.float32, nout).copy()
    return r


class Intonation(C.Structure):
    """trm_intonation (include/trm_c_api.h)."""
    _fields_ = [("useMicroIntonation", C.c_int32), ("useMacroIntonation", C.c_int32), ("useSmoothIntonation", C.c_int32),
                ("useDrift", C.c_int32), ("driftDeviation", C.c_float), ("driftCutoff", C.c_float), ("pitchMean", C.c_double),
                ("timeQuantization", C.c_uint32), ("startTime_ms", C.c_uint32), ("endTime_ms", C.c_uint32), ("driftSeed", C.c_float)]


def generate_frames(times, values, settings):
    """EventList.m:883-1061 restated (oracle/evt_oracle.c): (times u32[n], values f64[n,36]) -> frames f32[m,16]."""
    t = np.ascontiguousarray(times, dtype=np.uint32)
    v = np.ascontiguousarray(values, dtype=np.float64).reshape(-1, 36)
    s = Intonation.from_buffer_copy(bytes(settings)) if not isinstance(settings, Intonation) else settings
    n = C.c_size_t()
    lib().trm_oracle_count_frames(t.ctypes.data, len(t), C.addressof(s), C.byref(n))
    out = np.zeros((max(n.value, 1), 16), dtype=np.float32)
    m = C.c_size_t()
    lib().trm_oracle_generate_frames(t.ctypes.data, v.ctypes.data, len(t), C.addressof(s), out.ctypes.data, n.value, C.byref(m))
    assert m.value == n.value
    return out[:m.value]
