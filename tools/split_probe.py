#!/usr/bin/env python3
"""Time-split launches timed against whole utterances (device-resident batches; kernel time by hipEvents incl. the pre-pass).
usage: split_probe.py [config3|static|tv] ...   (default: all)
  config3: BASELINE configs[3], 1024 ragged utterances (sorted longest first)
  static:  configs[1]-style batches of 1024 .. 16384 static vowels x 1 s
  tv:      8192 time-varying voices x 1 s (configs[4]'s per-GPU shard)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import cases
import gnuspeech_amd as g

pd = cases.monet_default_params(44100.0)


def timed(frames, label, settings, reps=5, kernel="auto"):
    b = g.TRMBatch(g.TRMInputParameters.from_dict(pd))
    st = b.prepare_device(frames)
    for sp in settings:
        b.set_kernel(kernel)
        b.set_time_split(sp)
        b.synthesize_device(st); torch.cuda.synchronize()
        b.kernel_time_ms()
        t0 = time.perf_counter()
        for _ in range(reps):
            b.synthesize_device(st)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        km, n = b.kernel_time_ms()
        print("%-52s split %-5s -> %-9s %s  %.3f ms wall, %.3f ms device, %.3e samples/s" % (
            label, sp, b.last_time_split, b.last_kernel, dt * 1e3, km / max(1, n), st["total_out"] / dt), flush=True)


AUTO_ONLY = "--auto" in sys.argv
which = [x for x in sys.argv[1:] if not x.startswith("--")] or ["config3", "static", "tv"]
if AUTO_ONLY:
    _timed = timed
    timed = lambda fr, label, settings, **k: _timed(fr, label, ["off", "auto"], **k)
if "config3" in which:
    utt = cases.config4_frames(1024)
    utt.sort(key=len, reverse=True)
    timed(utt, "configs[3] 1024 ragged utterances", ["off", "auto", 140, 105, 70, 50, 35])
if "static" in which:
    for V in ((64, 256, 1024, 2048, 4096, 6144, 8192, 10240, 12288, 14336, 16384, 24576, 32768) if AUTO_ONLY else (1024, 4096, 8192, 12288, 16384)):
        fr = cases.config2_frames(V, nframes=251)
        timed(fr, "%d static vowels x 1 s" % V, ["off", "auto", 125, 84, 63, 50, 42, 32, 25])
if "tv" in which:
    fr = cases.config3_frames(8192, nframes=251)
    timed(fr, "8192 time-varying voices x 1 s", ["off", "auto", 125, 63, 42, 32])
