#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-buffer entry (trm_batch_synthesize_host) on the bench workload; DESIGN.md quotes it
next to bench.py's device-resident `value` (it is never `value`)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import cases
import gnuspeech_amd as g
V = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
fr = cases.config2_frames(V, nframes=251)
b = g.TRMBatch(g.TRMInputParameters.from_dict(cases.monet_default_params(44100.0)))
voices = list(fr)
for name, arg, reuse in (("list of voices, fresh output buffer", voices, False), ("list of voices, kept output buffer", voices, True),
                         ("[V,N,16] array, kept output buffer", np.ascontiguousarray(fr, dtype=np.float32), True)):
    pcm = None                                      # (drop the previous case's 700 MB buffer outside the timed region)
    b.synthesize(arg, reuse_output=reuse)
    t0 = time.perf_counter()
    pcm, ns, mx = b.synthesize(arg, reuse_output=reuse)
    dt = time.perf_counter() - t0
    print("host-buffer entry, %d voices x 1 s, %s: %.1f ms -> %.3e samples/s (frame packing + H2D + kernel + D2H fp32 PCM)" % (
        V, name, dt * 1e3, int(ns.sum()) / dt))
arr = np.ascontiguousarray(fr, dtype=np.float32)
b.synthesize_int16(arr, reuse_output=True)
t0 = time.perf_counter()
pcm, ns, mx = b.synthesize_int16(arr, reuse_output=True)
dt = time.perf_counter() - t0
print("host-buffer entry, %d voices x 1 s, [V,N,16] array, kept output buffer, int16 PCM out (scaled on the device): %.1f ms -> %.3e samples/s" % (
    V, dt * 1e3, int(ns.sum()) / dt))
