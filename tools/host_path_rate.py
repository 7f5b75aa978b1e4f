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
b.synthesize(voices[:64])
t0 = time.perf_counter()
pcm, ns, mx = b.synthesize(voices)
dt = time.perf_counter() - t0
print("host-buffer entry, %d voices x 1 s: %.1f ms -> %.3e samples/s (H2D frames + kernel + D2H fp32 PCM + per-voice copies)" % (V, dt * 1e3, int(ns.sum()) / dt))
