#!/usr/bin/env python3
"""Diagnostic: parity of both kernel forms at unusual tube lengths / rate ratios."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import cases, oracle_lib as O
import gnuspeech_amd as g
rows = cases.load_gnuspeech_rows()
for length, rate in ((30.0, 44100.0), (24.0, 44100.0), (20.0, 44100.0), (15.8, 22050.0)):
    pd = cases.monet_default_params(rate); pd["length"] = length
    voices = [rows[i:i + 40].copy() for i in range(0, 200, 11)]
    op = O.InputParams.from_dict(pd)
    ref = [O.synthesize(op, np.asarray(v, np.float32).astype(np.float64)) for v in voices]
    for form in ("wide", "quad"):
        b = g.TRMBatch(g.TRMInputParameters.from_dict(pd)); b.set_kernel(form)
        pcm, ns, mx = b.synthesize(voices)
        errs = []
        for v, o in enumerate(ref):
            e = (pcm[v].astype(np.float64) - o["samples"]) / o["maximumSampleValue"]
            errs.append(float(np.sqrt(np.mean(e * e))))
            if errs[-1] > 1e-4 and v == 0:
                bad = np.nonzero(np.abs(e) > 1e-3)[0]
                print("   first bad output index", bad[:5], "of", len(e))
        print("length %.1f rate %.0f %s: derived sr %d cp %d inc %d  worst rms %.3e (voice %d) n=%d" % (
            length, rate, form, b.derived["sampleRate"], b.derived["controlPeriod"], b.derived["timeRegisterIncrement"], max(errs), int(np.argmax(errs)), ns[0]))
