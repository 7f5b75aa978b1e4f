#!/usr/bin/env python3
"""The other BASELINE.json configurations, timed for the record (profiles/configs_r01.txt); bench.py's contract line
stays configs[1].
  configs[2]  4096 time-varying tubes x 1 s (gnuspeech.input tracks), device-resident
  configs[3]  1024 ragged utterances (0.6 - 6 s) end to end through the host-buffer entry (PCIe-inclusive), and the
              same batch device-resident, sorted by length like gnuspeech_amd/shard.py does
"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import cases
import gnuspeech_amd as g

pd = cases.monet_default_params(44100.0)
b = g.TRMBatch(g.TRMInputParameters.from_dict(pd))


def timed_device(frames, label, reps=3):
    st = b.prepare_device(frames)
    b.synthesize_device(st); torch.cuda.synchronize()
    b.kernel_time_ms()
    t0 = time.perf_counter()
    for _ in range(reps):
        b.synthesize_device(st)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print("%s: %.2f ms per pass, %.3e samples/s (%s form), %d output samples" % (label, dt * 1e3, st["total_out"] / dt, b.last_kernel, st["total_out"]))


fr = cases.config3_frames(4096, nframes=251)
timed_device(fr, "configs[2] 4096 time-varying tubes x 1 s, device-resident")
utt = cases.config4_frames(1024)
caller_order = list(utt)
utt.sort(key=len)
timed_device(utt, "configs[3] 1024 ragged utterances (%.1f - %.1f s, %.0f s of speech), device-resident" % (len(utt[0]) / 250, len(utt[-1]) / 250, sum(len(u) - 1 for u in utt) / 250))
# the host-buffer entry takes the batch in the CALLER's order (random lengths): the library orders the voices by length itself
for label, fn, kw in (("fresh fp32 output buffer", b.synthesize, {}), ("kept fp32 output buffer", b.synthesize, {"reuse_output": True}),
                      ("kept int16 output buffer (scaled on the device)", b.synthesize_int16, {"reuse_output": True})):
    fn(caller_order, **kw)
    t0 = time.perf_counter()
    pcm, ns, mx = fn(caller_order, **kw)
    dt = time.perf_counter() - t0
    print("configs[3] same batch, caller order, through the host-buffer entry (packing + H2D + kernel + D2H), %s: %.1f ms, %.3e samples/s" % (label, dt * 1e3, int(ns.sum()) / dt))
    pcm = None
