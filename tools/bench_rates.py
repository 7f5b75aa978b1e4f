#!/usr/bin/env python3
"""Device-resident time of the bench batch (4096 static voices x 1 s) at other output rates: 44.1 kHz takes the converter's
up-sampling branch inside the tube kernel, 22.05 / 16 / 8 kHz (tube rate 19 750 Hz above the output rate) the down-sampling
branch: tube-rate samples through HBM + trm_downsample_kernel.  For profiles/; bench.py's contract line stays 44.1 kHz."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import cases
import gnuspeech_amd as g

V = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
fr = cases.config2_frames(V, nframes=251)
for rate in (44100.0, 48000.0, 22050.0, 16000.0, 8000.0):
    pd = cases.monet_default_params(rate)
    b = g.TRMBatch(g.TRMInputParameters.from_dict(pd))
    st = b.prepare_device(fr)
    b.synthesize_device(st); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        b.synthesize_device(st)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    print("output rate %6.0f Hz (tube %d Hz): %.2f ms per pass, %.3e output samples/s, %.3e tube samples/s (%s form)" % (
        rate, b.derived["sampleRate"], dt * 1e3, st["total_out"] / dt, V * 250 * b.derived["controlPeriod"] / dt, b.last_kernel))
for crate in (100.0, 500.0, 1000.0):
    pd = cases.monet_default_params(44100.0)
    pd["controlRate"] = crate
    b = g.TRMBatch(g.TRMInputParameters.from_dict(pd))
    nfr = int(crate) + 1
    frc = np.repeat(fr[:, :1, :], nfr, axis=1)                      # 1 s of the same static voices at this control rate
    st = b.prepare_device(frc)
    b.synthesize_device(st); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        b.synthesize_device(st)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    print("control rate %4.0f Hz (control period %d tube samples), 44.1 kHz: %.2f ms per pass, %.3e output samples/s (%s form)" % (
        crate, b.derived["controlPeriod"], dt * 1e3, st["total_out"] / dt, b.last_kernel))
