"""ctypes binding of libtrm_hip.so (include/trm_c_api.h).  The library is the product; this module
only marshals.  Import fails loudly if the shared object is missing: there is no CPU fallback."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# TRM_LIB selects a diagnostic build (tools/stage_profile.py); the product is libtrm_hip.so
LIB_PATH = os.environ.get("TRM_LIB") or os.path.join(_HERE, "libtrm_hip.so")

TRM_OK = 0
TRM_KERNEL_AUTO, TRM_KERNEL_WIDE, TRM_KERNEL_QUAD = 0, 1, 2
(TRM_EINVAL, TRM_EINVAL_LENGTH, TRM_EFIR, TRM_ENOMEM, TRM_EHIP, TRM_ENODEVICE, TRM_EIO, TRM_EPARSE,
 TRM_ESILENT, TRM_ERANGE) = range(1, 11)


class TrmInputParams(C.Structure):
    """trm_input_params == TRMInputParameters (Frameworks/Tube/TRMInputParameters.h:26-54)."""
    _fields_ = [
        ("outputFileFormat", C.c_int32), ("outputRate", C.c_float), ("controlRate", C.c_float),
        ("volume", C.c_double), ("channels", C.c_int32), ("balance", C.c_double),
        ("waveform", C.c_int32), ("tp", C.c_double), ("tnMin", C.c_double), ("tnMax", C.c_double),
        ("breathiness", C.c_double), ("length", C.c_double), ("temperature", C.c_double),
        ("lossFactor", C.c_double), ("apScale", C.c_double), ("mouthCoef", C.c_double),
        ("noseCoef", C.c_double), ("noseRadius", C.c_double * 6), ("throatCutoff", C.c_double),
        ("throatVol", C.c_double), ("usesModulation", C.c_int32), ("mixOffset", C.c_double),
    ]


class TrmParameters(C.Structure):
    """trm_parameters == TRMParameters (Frameworks/Tube/TRMParameters.h:9-17), 16 doubles."""
    _fields_ = [
        ("glottalPitch", C.c_double), ("glottalVolume", C.c_double), ("aspirationVolume", C.c_double),
        ("fricationVolume", C.c_double), ("fricationPosition", C.c_double),
        ("fricationCenterFrequency", C.c_double), ("fricationBandwidth", C.c_double),
        ("radius", C.c_double * 8), ("velum", C.c_double),
    ]


class TrmIntonation(C.Structure):
    """trm_intonation (include/trm_c_api.h): MMIntonation's switches + pitch mean + time range."""
    _fields_ = [("useMicroIntonation", C.c_int32), ("useMacroIntonation", C.c_int32), ("useSmoothIntonation", C.c_int32),
                ("useDrift", C.c_int32), ("driftDeviation", C.c_float), ("driftCutoff", C.c_float), ("pitchMean", C.c_double),
                ("timeQuantization", C.c_uint32), ("startTime_ms", C.c_uint32), ("endTime_ms", C.c_uint32), ("driftSeed", C.c_float)]


class TrmDerived(C.Structure):
    _fields_ = [
        ("controlPeriod", C.c_int32), ("sampleRate", C.c_int32), ("actualTubeLength", C.c_double),
        ("sampleRateRatio", C.c_double), ("timeRegisterIncrement", C.c_uint32),
        ("phaseIncrement", C.c_uint32), ("padSize", C.c_int32), ("firTaps", C.c_int32),
    ]


# every symbol include/trm_c_api.h declares
EXPORTS = [
    "trm_strerror", "trm_last_error", "trm_data_list_read_file", "trm_data_list_write_file", "trm_free",
    "trm_tube_create", "trm_tube_destroy", "trm_tube_derived", "trm_tube_print_input_data", "trm_tube_synthesize",
    "trm_tube_number_samples", "trm_tube_maximum_sample_value", "trm_tube_samples",
    "trm_tube_save_output_to_file", "trm_tube_generate_wav_data", "trm_write_sound_file",
    "trm_batch_create", "trm_batch_destroy", "trm_batch_derived", "trm_batch_samples_for_frames",
    "trm_derive", "trm_samples_for_frames",
    "trm_batch_synthesize_host", "trm_batch_synthesize_host_int16", "trm_batch_synthesize_device", "trm_batch_scale_to_int16_device",
    "trm_sound_file_size", "trm_batch_sound_files_device",
    "trm_shard_voices", "trm_multi_create", "trm_multi_destroy", "trm_multi_synthesize_host", "trm_multi_synthesize_host_int16",
    "trm_stream_create", "trm_stream_destroy", "trm_stream_samples_for_push", "trm_stream_samples_for_finish",
    "trm_stream_push", "trm_stream_finish", "trm_stream_set_mode", "trm_stream_mode", "trm_stream_set_slice", "trm_stream_slice", "trm_stream_push_device", "trm_stream_finish_device", "trm_stream_kernel",
    "trm_events_count_frames", "trm_drift_seed_after", "trm_batch_generate_frames_device", "trm_batch_generate_frames_host",
    "trm_batch_set_kernel", "trm_batch_last_kernel", "trm_batch_set_time_split", "trm_batch_last_time_split", "trm_batch_hint_frames",
    "trm_batch_kernel_time_ms", "trm_batch_set_timing", "trm_batch_noise_table", "trm_device_count", "trm_build_info", "trm_kernel_blocks_per_cu", "trm_kernel_blocks_per_cu_form",
]

_lib = None


class TrmError(RuntimeError):
    def __init__(self, code, detail):
        super().__init__("libtrm_hip: %s (code %d): %s" % (lib().trm_strerror(code).decode(), code, detail))
        self.code = code


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError("%s is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(make -C gnuspeech_amd/csrc).  There is no CPU fallback." % LIB_PATH)
    # torch (the owner of device buffers / streams in this package) bundles its own HIP runtime under
    # the same SONAME as /opt/rocm's.  Whichever is loaded first serves the whole process, and torch
    # cannot run on the system copy: so when torch is installed, let it load its runtime first.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(LIB_PATH)
    vp = C.c_void_p
    L.trm_strerror.argtypes = [C.c_int]
    L.trm_strerror.restype = C.c_char_p
    L.trm_last_error.restype = C.c_char_p
    L.trm_build_info.restype = C.c_char_p
    L.trm_device_count.restype = C.c_int
    L.trm_free.argtypes = [vp]
    L.trm_data_list_read_file.argtypes = [C.c_char_p, C.POINTER(TrmInputParams),
                                          C.POINTER(C.POINTER(TrmParameters)), C.POINTER(C.c_size_t)]
    L.trm_data_list_write_file.argtypes = [C.c_char_p, C.POINTER(TrmInputParams), C.POINTER(TrmParameters), C.c_size_t]
    L.trm_tube_create.argtypes = [C.POINTER(TrmInputParams), C.c_int, C.POINTER(vp)]
    L.trm_tube_destroy.argtypes = [vp]
    L.trm_tube_derived.argtypes = [vp, C.POINTER(TrmDerived)]
    L.trm_tube_synthesize.argtypes = [vp, C.POINTER(TrmParameters), C.c_size_t]
    L.trm_tube_print_input_data.argtypes = [vp, C.POINTER(TrmParameters), C.c_size_t]
    L.trm_tube_number_samples.argtypes = [vp]
    L.trm_tube_number_samples.restype = C.c_size_t
    L.trm_tube_maximum_sample_value.argtypes = [vp]
    L.trm_tube_maximum_sample_value.restype = C.c_double
    L.trm_tube_samples.argtypes = [vp]
    L.trm_tube_samples.restype = C.POINTER(C.c_float)
    L.trm_tube_save_output_to_file.argtypes = [vp, C.c_char_p]
    L.trm_tube_generate_wav_data.argtypes = [vp, vp, C.c_size_t, C.POINTER(C.c_size_t)]
    L.trm_batch_create.argtypes = [C.POINTER(TrmInputParams), C.c_int, C.POINTER(vp)]
    L.trm_batch_destroy.argtypes = [vp]
    L.trm_batch_derived.argtypes = [vp, C.POINTER(TrmDerived)]
    L.trm_batch_samples_for_frames.argtypes = [vp, C.c_size_t]
    L.trm_batch_samples_for_frames.restype = C.c_size_t
    L.trm_derive.argtypes = [C.POINTER(TrmInputParams), C.POINTER(TrmDerived)]
    L.trm_samples_for_frames.argtypes = [C.POINTER(TrmInputParams), C.c_size_t]
    L.trm_samples_for_frames.restype = C.c_size_t
    L.trm_batch_synthesize_host.argtypes = [vp, C.c_size_t, vp, vp, vp, vp, vp, vp, vp]
    L.trm_batch_synthesize_host_int16.argtypes = [vp, C.c_size_t, vp, vp, vp, vp, vp, vp, vp, C.c_int]
    L.trm_shard_voices.argtypes = [vp, C.c_size_t, C.c_size_t, vp]
    L.trm_multi_create.argtypes = [C.POINTER(TrmInputParams), vp, C.c_size_t, C.POINTER(vp)]
    L.trm_multi_destroy.argtypes = [vp]
    L.trm_multi_synthesize_host.argtypes = [vp, C.c_size_t, vp, vp, vp, vp, vp, vp, vp]
    L.trm_multi_synthesize_host_int16.argtypes = [vp, C.c_size_t, vp, vp, vp, vp, vp, vp, vp, C.c_int]
    L.trm_batch_synthesize_device.argtypes = [vp, C.c_size_t, vp, vp, vp, C.c_uint32, vp, vp, vp, vp, vp]
    L.trm_batch_scale_to_int16_device.argtypes = [vp, C.c_size_t, vp, vp, vp, vp, vp, C.c_int, vp]
    L.trm_sound_file_size.argtypes = [C.POINTER(TrmInputParams), C.c_size_t]
    L.trm_sound_file_size.restype = C.c_size_t
    L.trm_batch_sound_files_device.argtypes = [vp, C.c_size_t, vp, vp, vp, vp, vp, vp, vp]
    L.trm_batch_kernel_time_ms.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_uint32)]
    L.trm_stream_create.argtypes = [C.POINTER(TrmInputParams), C.c_int, C.c_size_t, C.POINTER(vp)]
    L.trm_stream_destroy.argtypes = [vp]
    L.trm_stream_samples_for_push.argtypes = [vp, C.c_size_t]
    L.trm_stream_samples_for_push.restype = C.c_size_t
    L.trm_stream_samples_for_finish.argtypes = [vp]
    L.trm_stream_samples_for_finish.restype = C.c_size_t
    L.trm_stream_push.argtypes = [vp, vp, C.c_size_t, vp, C.c_size_t, C.POINTER(C.c_uint32), vp]
    L.trm_stream_finish.argtypes = [vp, vp, C.c_size_t, C.POINTER(C.c_uint32), vp]
    L.trm_stream_push_device.argtypes = [vp, vp, C.c_size_t, vp, C.c_size_t, C.POINTER(C.c_uint32), vp, vp]
    L.trm_stream_finish_device.argtypes = [vp, vp, C.c_size_t, C.POINTER(C.c_uint32), vp, vp]
    L.trm_stream_kernel.argtypes = [vp]
    L.trm_stream_set_mode.argtypes = [vp, C.c_int]
    L.trm_stream_mode.argtypes = [vp]
    L.trm_stream_set_slice.argtypes = [vp, C.c_uint32]
    L.trm_stream_slice.argtypes = [vp]
    L.trm_stream_slice.restype = C.c_uint32
    L.trm_events_count_frames.argtypes = [vp, C.c_size_t, C.POINTER(TrmIntonation), C.POINTER(C.c_size_t)]
    L.trm_drift_seed_after.argtypes = [C.c_float, C.c_size_t]
    L.trm_drift_seed_after.restype = C.c_float
    L.trm_batch_generate_frames_device.argtypes = [vp, C.c_size_t, vp, vp, vp, vp, C.POINTER(TrmIntonation), vp, vp, vp, vp]
    L.trm_batch_generate_frames_host.argtypes = [vp, vp, vp, C.c_size_t, C.POINTER(TrmIntonation), vp, C.c_size_t, C.POINTER(C.c_size_t)]
    L.trm_write_sound_file.argtypes = [C.POINTER(TrmInputParams), vp, C.c_size_t, C.c_float, C.c_char_p]
    L.trm_batch_set_kernel.argtypes = [vp, C.c_int]
    L.trm_batch_set_timing.argtypes = [vp, C.c_int]
    L.trm_kernel_blocks_per_cu_form.argtypes = [C.c_int]
    L.trm_batch_last_kernel.argtypes = [vp]
    L.trm_batch_set_time_split.argtypes = [vp, C.c_int]
    L.trm_batch_last_time_split.argtypes = [vp, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    L.trm_batch_hint_frames.argtypes = [vp, vp, C.c_size_t]
    L.trm_batch_noise_table.argtypes = [vp, vp, C.c_size_t]
    for name in EXPORTS:
        getattr(L, name)
    _lib = L
    return L


def check(rc):
    if rc != TRM_OK:
        raise TrmError(rc, lib().trm_last_error().decode(errors="replace"))
