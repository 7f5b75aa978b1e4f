#!/usr/bin/env python3
"""Diagnostic: both kernel forms vs the oracle over a sweep of tube lengths at high output rates (rate ratios 2 - 8)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, cases, gnuspeech_amd as g, oracle_lib as O
rows = cases.load_gnuspeech_rows()
voices = [rows[:200].copy(), rows[50:343].copy(), np.concatenate([rows, rows])[:600].copy()]
for rate in [float(x) for x in os.environ.get("SCAN_RATES", "96000,48000,64000").split(",")]:
  for L in np.arange(float(os.environ.get("SCAN_L0", "14")), 30.5, float(os.environ.get("SCAN_DL", "1"))):
    pd = cases.monet_default_params(rate); pd["length"] = float(L)
    ip = g.TRMInputParameters.from_dict(pd); op = O.InputParams.from_dict(pd)
    ref = [O.synthesize(op, np.asarray(v, np.float32).astype(np.float64)) for v in voices]
    res = []
    for form in ("quad", "wide"):
        b = g.TRMBatch(ip); b.set_kernel(form)
        pcm, ns, mx = b.synthesize(voices)
        worst = 0.0
        for v, o in enumerate(ref):
            if int(ns[v]) != o["numberSamples"]: worst = 9.0; continue
            e = (pcm[v].astype(np.float64) - o["samples"]) / o["maximumSampleValue"]
            worst = max(worst, float(np.sqrt(np.mean(e * e))))
        res.append(worst)
    print("rate %.0f length %.1f ratio %.2f: quad %.1e wide %.1e %s" % (rate, L, rate / b.derived["sampleRate"], res[0], res[1], "  <-- FAIL" if max(res) > 1e-5 else ""))
