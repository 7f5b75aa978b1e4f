"""Streaming synthesis over the C ABI: an utterance delivered in chunks of control frames, PCM
returned per chunk; the concatenation equals one-shot synthesis bit for bit.  See include/trm_c_api.h
(trm_stream_*); what TRAcT's real-time loop (Applications/TRAcT/tube.c:1096-1190) maps onto."""
import ctypes as C

import numpy as np

from ._capi import check, lib


MODES = {"framework": 0, "tract": 1}      # TRM_STREAM_MODE_* (include/trm_c_api.h)


class TRMStream:
    def __init__(self, inputParameters, nvoices=1, device=-1, mode="framework"):
        self._h = C.c_void_p()
        self.nvoices = int(nvoices)
        self.inputParameters = inputParameters
        check(lib().trm_stream_create(C.byref(inputParameters.c), device, self.nvoices, C.byref(self._h)))
        if mode != "framework":
            self.set_mode(mode)

    @property
    def kernel(self):
        """"wide" (one voice per lane) or "quad" (four lanes per voice): fixed when the stream was created."""
        return {1: "wide", 2: "quad"}[lib().trm_stream_kernel(self._h)]

    def set_mode(self, mode):
        """"framework": Frameworks/Tube's loop (interpolated control periods); "tract": Applications/TRAcT/tube.c's own
        (every frame one control period of held parameters, x10 frication taps, x100 output; tube.c:1096-1190, 1371)."""
        check(lib().trm_stream_set_mode(self._h, MODES[mode]))

    def set_slice(self, tube_samples):
        """"tract" mode only: the tube samples one pushed frame stands for (0 = a control period); see trm_stream_set_slice."""
        check(lib().trm_stream_set_slice(self._h, int(tube_samples)))

    @property
    def slice(self):
        return lib().trm_stream_slice(self._h)

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            try:
                lib().trm_stream_destroy(h)
            except Exception:
                pass
            self._h = None

    def push(self, frames):
        """frames: [nvoices, n, 16] (or [n, 16] for one voice).  Returns (pcm [nvoices, m] float32, max |sample| [nvoices])."""
        f = np.ascontiguousarray(frames, dtype=np.float32)
        if f.ndim == 2:
            f = f[None]
        assert f.shape[0] == self.nvoices and f.shape[2] == 16
        m = lib().trm_stream_samples_for_push(self._h, f.shape[1])
        return self._run(lambda out, n, mx: lib().trm_stream_push(self._h, f.ctypes.data, f.shape[1], out, max(m, 1), n, mx), m)

    def finish(self):
        m = lib().trm_stream_samples_for_finish(self._h)
        return self._run(lambda out, n, mx: lib().trm_stream_finish(self._h, out, max(m, 1), n, mx), m)

    def _run(self, call, m):
        out = np.zeros((self.nvoices, max(m, 1)), dtype=np.float32)
        mx = np.zeros(self.nvoices, dtype=np.float32)
        n = C.c_uint32()
        check(call(out.ctypes.data, C.byref(n), mx.ctypes.data))
        assert n.value == m
        return out[:, :m], mx

    # ---- device-buffer forms (torch tensors on the stream's device; asynchronous on torch's current stream)
    def push_device(self, frames, out=None, max_out=None):
        """frames: float32 CUDA tensor [nvoices, n, 16].  Returns (pcm [nvoices, m] view of `out`, m): nothing crosses PCIe,
        nothing is waited for.  `out` (optional) = a float32 CUDA tensor [nvoices, pitch >= m] to write into."""
        import torch
        assert frames.is_cuda and frames.dtype == torch.float32 and frames.is_contiguous() and frames.shape[0] == self.nvoices and frames.shape[2] == 16
        m = lib().trm_stream_samples_for_push(self._h, frames.shape[1])
        return self._run_device(lambda o, pitch, n, mx, st: lib().trm_stream_push_device(self._h, frames.data_ptr(), frames.shape[1], o, pitch, n, mx, st),
                                m, frames.device, out, max_out)

    def finish_device(self, device=None, out=None, max_out=None):
        import torch
        m = lib().trm_stream_samples_for_finish(self._h)
        dev = out.device if out is not None else (device if device is not None else torch.device("cuda", torch.cuda.current_device()))
        return self._run_device(lambda o, pitch, n, mx, st: lib().trm_stream_finish_device(self._h, o, pitch, n, mx, st), m, dev, out, max_out)

    def _run_device(self, call, m, device, out, max_out):
        import torch
        if out is None:
            out = torch.empty((self.nvoices, max(m, 1)), dtype=torch.float32, device=device)
        assert out.is_cuda and out.dtype == torch.float32 and out.shape[0] == self.nvoices and out.stride(1) == 1 and out.shape[1] >= m
        n = C.c_uint32()
        st = torch.cuda.current_stream(device).cuda_stream
        check(call(out.data_ptr(), out.stride(0), C.byref(n), max_out.data_ptr() if max_out is not None else None, st))
        assert n.value == m
        return out[:, :m], m
