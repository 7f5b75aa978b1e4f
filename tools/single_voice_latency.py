#!/usr/bin/env python3
"""Latency of the reference-shaped single-utterance path (TRMTubeModel -synthesize through the C ABI): what a
TRMSynthesizer caller sees per utterance."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import cases
import gnuspeech_amd as g
rows = cases.load_gnuspeech_rows()
for nfr in (26, 251, 1001):
    dl = g.TRMDataList()
    dl.inputParameters = g.TRMInputParameters.from_dict(cases.monet_default_params(44100.0))
    fr = np.tile(rows, (nfr // len(rows) + 1, 1))[:nfr]
    dl.values = [g.TRMParameters(r) for r in fr]
    tube = g.TRMTubeModel.initWithInputData(dl)
    tube.synthesize()
    t = []
    for _ in range(10):
        t0 = time.perf_counter(); tube.synthesize(); t.append(time.perf_counter() - t0)
    print("%4d frames (%.1f s of speech): synthesize() %.2f ms median, %d samples" % (nfr, (nfr - 1) / 250.0, 1e3 * np.median(t), tube.numberSamples))
# what the reference's callers do per utterance (TRMSynthesizer.m:118-136): a FRESH tube, synthesize, discard
for rate in (44100.0, 16000.0):
    dl = g.TRMDataList()
    dl.inputParameters = g.TRMInputParameters.from_dict(cases.monet_default_params(rate))
    dl.values = [g.TRMParameters(r) for r in rows[:251]]
    t_init, t_syn = [], []
    for _ in range(6):
        t0 = time.perf_counter(); tube = g.TRMTubeModel.initWithInputData(dl); t1 = time.perf_counter()
        tube.synthesize(); t2 = time.perf_counter()
        t_init.append(t1 - t0); t_syn.append(t2 - t1)
        del tube
    print("fresh tube per utterance, %5.0f Hz: initWithInputData %.2f ms, first synthesize() of 1 s %.2f ms (medians of 6)" % (
        rate, 1e3 * np.median(t_init), 1e3 * np.median(t_syn)))
