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

    def set_mode(self, mode):
        """"framework": Frameworks/Tube's loop (interpolated control periods); "tract": Applications/TRAcT/tube.c's own
        (every frame one control period of held parameters, x10 frication taps, x100 output; tube.c:1096-1190, 1371)."""
        check(lib().trm_stream_set_mode(self._h, MODES[mode]))

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
