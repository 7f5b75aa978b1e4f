#!/usr/bin/env python3
"""Per-push latency of a small real-time stream through the HOST entry (trm_stream_push: frames in, PCM out, H2D + kernel + D2H):
what an interactive caller in TRAcT's place sees (Applications/TRAcT/tube.c:1096-1190 keeps a producer a little ahead of the audio
callback).  N voices, pushes of `chunk` control frames (4 ms of audio each at 250 Hz), median / p99 / max over `pushes` pushes.
usage: stream_latency.py [--mode framework|tract]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import cases
import gnuspeech_amd as g
mode = sys.argv[sys.argv.index("--mode") + 1] if "--mode" in sys.argv else "framework"
P = g.TRMInputParameters.from_dict(cases.monet_default_params(44100.0))
for N in (1, 16, 256, 4096):
    for chunk in (1, 5, 25):
        pushes = 200 if chunk < 25 else 80
        fr = cases.config3_frames(N, nframes=chunk * pushes + 1)
        s = g.TRMStream(P, nvoices=N, mode=mode)
        s.push(fr[:, :1])                                     # the first frame: no audio yet
        t = []
        for i in range(pushes):
            a = time.perf_counter()
            s.push(fr[:, 1 + i * chunk:1 + (i + 1) * chunk])
            t.append((time.perf_counter() - a) * 1e3)
        s.finish()
        t = np.array(t[5:])
        print("%5d voice(s), pushes of %2d frame(s) = %5.1f ms of audio (%s form): median %.3f ms, p99 %.3f, max %.3f  -> %.0fx real time" % (
            N, chunk, chunk * 4.0, s.kernel, np.median(t), np.percentile(t, 99), t.max(), chunk * 4.0 / np.median(t)))
