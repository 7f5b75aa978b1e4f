#!/usr/bin/env python3
"""Diagnostic: one seed of tests/test_gpu_parity.py::test_random_voices_and_tracks, with where the error starts."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import cases, oracle_lib as O
import gnuspeech_amd as g
seed = int(sys.argv[1]); overrides = dict(kv.split("=") for kv in sys.argv[2:])
rng = np.random.default_rng(1000 + seed)
pd = cases.monet_default_params(float(rng.choice([22050.0, 44100.0])))
pd.update(controlRate=float(rng.choice([100.0, 250.0, 500.0, 1000.0])), waveform=int(rng.integers(0, 2)),
          tp=float(rng.uniform(20, 45)), tnMin=float(rng.uniform(8, 20)), breathiness=float(rng.uniform(0, 10)),
          length=float(rng.uniform(11.0, 24.0)), temperature=float(rng.uniform(25, 40)), lossFactor=float(rng.uniform(0.1, 3.0)),
          apScale=float(rng.uniform(1.5, 5.0)), mouthCoef=float(rng.uniform(2000, 6000)), noseCoef=float(rng.uniform(2000, 6000)),
          noseRadius=[0.0] + [float(x) for x in rng.uniform(0.5, 2.5, 5)], throatCutoff=float(rng.uniform(500, 3000)),
          throatVol=float(rng.uniform(0, 24)), usesModulation=int(rng.integers(0, 2)), mixOffset=float(rng.uniform(30, 60)))
pd["tnMax"] = pd["tnMin"] + float(rng.uniform(5, 20))
voices = []
for _ in range(5):
    n = int(rng.integers(2, 60)); knots = max(2, n // 8); t = np.linspace(0, knots - 1, n)
    def track(lo, hi): return np.interp(t, np.arange(knots), rng.uniform(lo, hi, knots))
    voices.append(np.stack([track(-10, 6), track(0, 60), track(0, 20), track(0, 40), track(0, 7), track(500, 5000), track(200, 2500)] + [track(0.05, 2.5) for _ in range(8)] + [track(0.0, 1.2)], axis=1))
for k, v in overrides.items():
    pd[k] = type(pd[k])(float(v))
op = O.InputParams.from_dict(pd)
for form in ("wide", "quad"):
    b = g.TRMBatch(g.TRMInputParameters.from_dict(pd)); b.set_kernel(form)
    pcm, ns, mx = b.synthesize(voices)
    for v, fr in enumerate(voices):
        o = O.synthesize(op, np.asarray(fr, np.float32).astype(np.float64))
        e = (pcm[v].astype(np.float64) - o["samples"]) / o["maximumSampleValue"]
        bad = np.nonzero(np.abs(e) > 1e-3)[0]
        print("%s voice %d frames %d: rms %.3e  cp %d  first bad output %s of %d (tube sample ~%s)" % (
            form, v, len(fr), np.sqrt(np.mean(e * e)), b.derived["controlPeriod"], bad[:1], len(e),
            (bad[:1] * b.derived["timeRegisterIncrement"] >> 16) if len(bad) else ""))
