"""Batch front end: V independent tubes sharing one TRMInputParameters.

torch is used only as the owner of device memory and streams (plumbing): the kernels are
libtrm_hip.so's.  Layout in HBM:
    frames   fp32 [sum nframes][16]      voice v owns rows frame_offset[v] .. +nframes[v]
    pcm      fp32 [sum nsamples]         voice v's output at out_offset[v]
"""
import ctypes as C

import numpy as np

from ._capi import TrmDerived, check, lib


class TRMBatch:
    def __init__(self, inputParameters, device=-1):
        self._h = C.c_void_p()
        self.inputParameters = inputParameters
        check(lib().trm_batch_create(C.byref(inputParameters.c), device, C.byref(self._h)))
        d = TrmDerived()
        check(lib().trm_batch_derived(self._h, C.byref(d)))
        self.derived = {k: getattr(d, k) for k, _ in TrmDerived._fields_}

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            try:
                lib().trm_batch_destroy(h)
            except Exception:      # interpreter shutdown: the process is going away anyway
                pass
            self._h = None

    def samples_for_frames(self, nframes):
        return lib().trm_batch_samples_for_frames(self._h, int(nframes))

    # -------------------------------------------------------------- host buffers (incl. H2D / D2H)
    def synthesize(self, voices, reuse_output=False):
        """voices: list of [n_v,16] arrays (ragged allowed) or one [V,N,16] array.  Returns (list of fp32 PCM
        arrays, numberSamples uint32[V], maximumSampleValue float32[V]).  reuse_output=True returns views into a
        buffer this object keeps and overwrites on the next call (a fresh 700 MB numpy buffer costs more in first-
        touch page faults than the D2H copy that fills it, profiles/host_path_r01.txt): copy what must outlive
        the next call."""
        return self._synthesize_host(lib().trm_batch_synthesize_host, voices, reuse_output)

    def synthesize_int16(self, voices, for_wav_data=False, reuse_output=False):
        """As synthesize(), returning the containers' int16 PCM (TRMTubeModel.m:370-389 / :515-540): per voice an
        int16 array [n] (mono) or [n, 2] (inputParameters.channels == 2); the scaling runs on the device and half the
        bytes cross PCIe."""
        ch = 2 if self.inputParameters.channels == 2 else 1
        return self._synthesize_host(lib().trm_batch_synthesize_host_int16, voices, reuse_output, dtype=np.int16, channels=ch,
                                     extra=(int(bool(for_wav_data)),))

    def _synthesize_host(self, entry, voices, reuse_output, dtype=np.float32, channels=1, extra=()):
        V = len(voices)
        if isinstance(voices, np.ndarray) and voices.ndim == 3:
            nfr = np.full(V, voices.shape[1], dtype=np.uint32)
            frames = np.ascontiguousarray(voices.reshape(-1, 16), dtype=np.float32)
            if not len(frames):
                frames = np.zeros((1, 16), dtype=np.float32)
        else:
            nfr = np.array([len(v) for v in voices], dtype=np.uint32)
            frames = None
        foff = np.zeros(max(1, V), dtype=np.uint64)
        if V > 1:
            foff[1:V] = np.cumsum(nfr[:-1], dtype=np.uint64)
        if frames is None:
            frames = np.zeros((max(1, int(nfr.sum())), 16), dtype=np.float32)
            for v, fr in enumerate(voices):
                if len(fr):
                    frames[int(foff[v]):int(foff[v]) + len(fr)] = np.asarray(fr, dtype=np.float32)
        lut = {int(n): self.samples_for_frames(int(n)) for n in np.unique(nfr)}
        nout = np.array([lut[int(n)] for n in nfr], dtype=np.uint64)
        ooff = np.zeros(max(1, V), dtype=np.uint64)
        if V > 1:
            ooff[1:V] = np.cumsum(nout[:-1], dtype=np.uint64)
        total = max(1, int(nout.sum()))
        total *= channels
        if reuse_output:
            nbytes = total * np.dtype(dtype).itemsize
            if getattr(self, "_host_out", None) is None or self._host_out.size < nbytes:
                self._host_out = np.ones(nbytes + nbytes // 8, dtype=np.uint8)      # ones: every page touched
            out = self._host_out[:nbytes].view(dtype)
        else:
            out = np.zeros(total, dtype=dtype)
        ns = np.zeros(max(1, V), dtype=np.uint32)
        mx = np.zeros(max(1, V), dtype=np.float32)
        nfr_c = np.ascontiguousarray(nfr if V else np.zeros(1, np.uint32))
        check(entry(self._h, V, frames.ctypes.data, foff.ctypes.data, nfr_c.ctypes.data,
                                              out.ctypes.data, ooff.ctypes.data, ns.ctypes.data, mx.ctypes.data, *extra))
        if channels == 2:
            out = out.reshape(-1, 2)
        pcm = [out[int(ooff[v]):int(ooff[v]) + int(ns[v])] for v in range(V)]
        return pcm, ns[:V], mx[:V]

    # -------------------------------------------------------------- device buffers (torch tensors)
    def prepare_device(self, frames, device="cuda"):
        """Upload a batch once.  frames: [V,N,16] array (uniform) or list of [n_v,16] (ragged)."""
        import torch
        if isinstance(frames, np.ndarray) and frames.ndim == 3:
            V, N = frames.shape[:2]
            nfr = np.full(V, N, dtype=np.int64)
            flat = np.ascontiguousarray(frames.reshape(V * N, 16), dtype=np.float32)
        else:
            V = len(frames)
            nfr = np.array([len(v) for v in frames], dtype=np.int64)
            flat = (np.concatenate([np.asarray(v, dtype=np.float32).reshape(-1, 16) for v in frames])
                    if V else np.zeros((0, 16), np.float32))
        foff = np.zeros(max(1, V), dtype=np.int64)
        if V > 1:
            foff[1:V] = np.cumsum(nfr[:-1])
        lut = {int(n): self.samples_for_frames(int(n)) for n in np.unique(nfr)}
        nout_v = np.array([lut[int(n)] for n in nfr], dtype=np.int64)
        # every voice's PCM starts on a 128-byte boundary: the converter stores rows of 32 samples, and a row
        # that straddles two cache lines costs two partial writes instead of one full line
        pitch_v = (nout_v + 31) // 32 * 32
        ooff = np.zeros(max(1, V), dtype=np.int64)
        if V > 1:
            ooff[1:V] = np.cumsum(pitch_v[:-1])
        dev = torch.device(device)
        return {
            "V": V, "max_nframes": int(nfr.max()) if V else 0, "total_out": int(nout_v.sum()),
            "out_alloc": int(pitch_v.sum()),
            "nout": nout_v, "out_offset_host": ooff, "nframes_host": nfr,
            "frames": torch.from_numpy(flat if flat.size else np.zeros((1, 16), np.float32)).to(dev),
            "frame_offset": torch.from_numpy(foff).to(dev),
            "nframes": torch.from_numpy(nfr.astype(np.int32) if V else np.zeros(1, np.int32)).to(dev),
            "out_offset": torch.from_numpy(ooff).to(dev),
            "out": torch.zeros(max(1, int(pitch_v.sum())), dtype=torch.float32, device=dev),
            "number_samples": torch.zeros(max(1, V), dtype=torch.int32, device=dev),
            "max_sample": torch.zeros(max(1, V), dtype=torch.float32, device=dev),
        }

    def prepare_events_device(self, event_lists, settings, device="cuda"):
        """Upload a batch of event lists (list of (times u32[n], values f64[n,36])); the frames buffer is sized
        by trm_events_count_frames and filled on the device by generate_frames_device().  The returned state
        is what synthesize_device() takes: event lists -> PCM without the frames crossing PCIe."""
        import torch
        V = len(event_lists)
        nev = np.array([len(t) for t, _ in event_lists], dtype=np.int64)
        times = np.concatenate([np.asarray(t, dtype=np.uint32) for t, _ in event_lists]) if V else np.zeros(0, np.uint32)
        values = (np.concatenate([np.asarray(v, dtype=np.float64).reshape(-1, 36) for _, v in event_lists])
                  if V else np.zeros((0, 36)))
        eoff = np.zeros(max(1, V), dtype=np.int64)
        if V > 1:
            eoff[1:V] = np.cumsum(nev[:-1])
        nfr = np.zeros(max(1, V), dtype=np.int64)
        for v, (t, _) in enumerate(event_lists):
            n = C.c_size_t()
            t32 = np.ascontiguousarray(t, dtype=np.uint32)
            check(lib().trm_events_count_frames(t32.ctypes.data, len(t32), C.byref(settings), C.byref(n)))
            nfr[v] = n.value
        st = self.prepare_device([np.zeros((int(n), 16), np.float32) for n in nfr[:V]], device=device)
        dev = torch.device(device)
        st["settings"] = settings
        st["event_times"] = torch.from_numpy(times.astype(np.int32) if times.size else np.zeros(1, np.int32)).to(dev)
        st["event_values"] = torch.from_numpy(values if values.size else np.zeros((1, 36))).to(dev)
        st["event_offset"] = torch.from_numpy(eoff).to(dev)
        st["nevents"] = torch.from_numpy(nev.astype(np.int32) if V else np.zeros(1, np.int32)).to(dev)
        st["nframes_generated"] = torch.zeros(max(1, V), dtype=torch.int32, device=dev)
        return st

    def generate_frames_device(self, st, stream=None):
        """trm_tracks_kernel over a resident batch of event lists: fills st["frames"]."""
        import torch
        s = stream if stream is not None else torch.cuda.current_stream()
        check(lib().trm_batch_generate_frames_device(
            self._h, st["V"], st["event_times"].data_ptr(), st["event_values"].data_ptr(), st["event_offset"].data_ptr(),
            st["nevents"].data_ptr(), C.byref(st["settings"]), st["frames"].data_ptr(), st["frame_offset"].data_ptr(),
            st["nframes_generated"].data_ptr(), C.c_void_p(s.cuda_stream)))

    def synthesize_device(self, st, stream=None):
        """One pass of the hot path over a resident batch; asynchronous on `stream`
        (default: torch's current stream)."""
        import torch
        s = stream if stream is not None else torch.cuda.current_stream()
        # the lengths as the host knows them: AUTO sizes a ragged batch's segments by the work it really holds (trm_batch_hint_frames)
        nfh = st.get("nframes_host")
        if nfh is not None and st["V"] > 0:
            nfh = np.ascontiguousarray(nfh, dtype=np.uint32)
            check(lib().trm_batch_hint_frames(self._h, nfh.ctypes.data, st["V"]))
        check(lib().trm_batch_synthesize_device(
            self._h, st["V"], st["frames"].data_ptr(), st["frame_offset"].data_ptr(), st["nframes"].data_ptr(),
            st["max_nframes"], st["out"].data_ptr(), st["out_offset"].data_ptr(), st["number_samples"].data_ptr(),
            st["max_sample"].data_ptr(), C.c_void_p(s.cuda_stream)))

    def scale_to_int16_device(self, st, for_wav_data=False, stream=None):
        import torch
        s = stream if stream is not None else torch.cuda.current_stream()
        ch = 2 if self.inputParameters.channels == 2 else 1
        pcm16 = torch.zeros(max(1, st["out_alloc"] * ch), dtype=torch.int16, device=st["out"].device)
        check(lib().trm_batch_scale_to_int16_device(
            self._h, st["V"], st["out"].data_ptr(), st["out_offset"].data_ptr(), st["number_samples"].data_ptr(),
            st["max_sample"].data_ptr(), pcm16.data_ptr(), int(for_wav_data), C.c_void_p(s.cuda_stream)))
        return pcm16

    def sound_files_device(self, st, stream=None):
        """The batch's sound files composed on the device (trm_batch_sound_files_device): returns (uint8 CUDA tensor holding
        every voice's file image -- header + int16 payload in the container's byte order --, byte offsets, sizes)."""
        import torch
        s = stream if stream is not None else torch.cuda.current_stream()
        sizes = np.array([lib().trm_sound_file_size(C.byref(self.inputParameters.c), int(n)) for n in st["nout"]], dtype=np.int64)
        pitch = (sizes + 63) // 64 * 64
        foff = np.zeros(max(1, st["V"]), dtype=np.int64)
        if st["V"] > 1:
            foff[1:st["V"]] = np.cumsum(pitch[:-1])
        files = torch.zeros(max(1, int(pitch.sum())), dtype=torch.uint8, device=st["out"].device)
        d_foff = torch.from_numpy(foff).to(st["out"].device)
        check(lib().trm_batch_sound_files_device(
            self._h, st["V"], st["out"].data_ptr(), st["out_offset"].data_ptr(), st["number_samples"].data_ptr(),
            st["max_sample"].data_ptr(), files.data_ptr(), d_foff.data_ptr(), C.c_void_p(s.cuda_stream)))
        return files, foff, sizes

    def noise_table(self, n):
        out = np.zeros(int(n), dtype=np.float32)
        check(lib().trm_batch_noise_table(self._h, out.ctypes.data, int(n)))
        return out

    def set_kernel(self, kernel):
        """'auto' | 'wide' (one voice per lane) | 'quad' (four lanes per voice) | 'oct' (eight); see include/trm_c_api.h."""
        check(lib().trm_batch_set_kernel(self._h, {"auto": 0, "wide": 1, "quad": 2, "oct": 3}[kernel]))

    def set_time_split(self, periods):
        """'auto' (default) | 'off' | control periods per segment: cut every utterance in time and run the segments side by
        side, each from rest a warm-up ahead (include/trm_c_api.h: trm_batch_set_time_split)."""
        check(lib().trm_batch_set_time_split(self._h, {"auto": -1, "off": 0}.get(periods, periods)))

    @property
    def last_time_split(self):
        """(control periods per segment, warm-up control periods) of the last launch; (0, 0) = whole utterances."""
        p, w = C.c_uint32(), C.c_uint32()
        check(lib().trm_batch_last_time_split(self._h, C.byref(p), C.byref(w)))
        return p.value, w.value

    def set_timing(self, on):
        """Launch timing on / off; off, synthesize_device is pure stream work and can be captured into a HIP graph."""
        check(lib().trm_batch_set_timing(self._h, int(bool(on))))

    @property
    def last_kernel(self):
        return {0: "auto", 1: "wide", 2: "quad", 3: "oct"}[lib().trm_batch_last_kernel(self._h)]

    def kernel_time_ms(self):
        t = C.c_double()
        n = C.c_uint32()
        check(lib().trm_batch_kernel_time_ms(self._h, C.byref(t), C.byref(n)))
        return t.value, n.value


def shard_voices(nframes, nshards):
    """Contiguous voice ranges balanced by frame count: bounds[g] .. bounds[g+1] is shard g (trm_shard_voices; what
    trm_multi_synthesize_host uses per device, and what a one-process-per-GPU launcher can use per rank)."""
    nfr = np.ascontiguousarray(nframes, dtype=np.uint32)
    bounds = np.zeros(int(nshards) + 1, dtype=np.uint64)
    assert bounds.itemsize == C.sizeof(C.c_size_t)
    check(lib().trm_shard_voices(nfr.ctypes.data if len(nfr) else None, len(nfr), int(nshards), bounds.ctypes.data))
    return [int(x) for x in bounds]


class TRMMultiBatch(TRMBatch):
    """One process, several GPUs: contiguous voice ranges per device, one host thread and stream per
    device, no collective.  Host-buffer entry only; `devices` may repeat a device."""

    def __init__(self, inputParameters, devices):
        self._h = C.c_void_p()
        self.inputParameters = inputParameters
        self.devices = [int(d) for d in devices]
        dv = (C.c_int * len(self.devices))(*self.devices)
        check(lib().trm_multi_create(C.byref(inputParameters.c), dv, len(self.devices), C.byref(self._h)))
        d = TrmDerived()
        check(lib().trm_derive(C.byref(inputParameters.c), C.byref(d)))
        self.derived = {k: getattr(d, k) for k, _ in TrmDerived._fields_}

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            try:
                lib().trm_multi_destroy(h)
            except Exception:
                pass
            self._h = None

    def samples_for_frames(self, nframes):
        return lib().trm_samples_for_frames(C.byref(self.inputParameters.c), int(nframes))

    def synthesize(self, voices, reuse_output=False):
        return self._synthesize_host(lib().trm_multi_synthesize_host, voices, reuse_output)

    def synthesize_int16(self, voices, for_wav_data=False, reuse_output=False):
        ch = 2 if self.inputParameters.channels == 2 else 1
        return self._synthesize_host(lib().trm_multi_synthesize_host_int16, voices, reuse_output, dtype=np.int16, channels=ch,
                                     extra=(int(bool(for_wav_data)),))

    def _device_only(self, *a, **k):
        raise NotImplementedError("TRMMultiBatch carries the host-buffer entries only (one trm_batch per device inside the library)")

    prepare_device = synthesize_device = scale_to_int16_device = prepare_events_device = generate_frames_device = _device_only
    sound_files_device = _device_only
    noise_table = set_kernel = kernel_time_ms = set_timing = set_time_split = _device_only
    last_kernel = last_time_split = property(_device_only)
