#!/usr/bin/env python3
"""Wider sweep of the randomized parity test (tests/test_gpu_parity.py::test_random_voices_and_tracks): more seeds,
longer and more voices.  Prints every (seed, form, voice) above tolerance.   usage: fuzz_parity.py first last [maxframes]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import cases, oracle_lib as O
import gnuspeech_amd as g
first, last = int(sys.argv[1]), int(sys.argv[2])
maxframes = int(sys.argv[3]) if len(sys.argv) > 3 else 300
broad = len(sys.argv) > 4 and sys.argv[4] == "broad"        # wider (still legal) parameter and track ranges
bad = 0; unstable = 0; floor = 0; worst = 0.0; worstBy = {}; allr = []
for seed in range(first, last):
    rng = np.random.default_rng(5000 + seed)
    pd = cases.monet_default_params(float(rng.choice([22050.0, 44100.0, 16000.0, 8000.0, 11025.0, 48000.0, 32000.0, 12000.0, 96000.0])))
    pd.update(controlRate=float(rng.choice([100.0, 250.0, 500.0, 1000.0])), waveform=int(rng.integers(0, 2)),
              tp=float(rng.uniform(20, 45)), tnMin=float(rng.uniform(8, 20)), breathiness=float(rng.uniform(0, 10)),
              length=float(rng.uniform(10.0, 30.0)), temperature=float(rng.uniform(25, 40)), lossFactor=float(rng.uniform(0.1, 3.0)),
              apScale=float(rng.uniform(1.5, 5.0)), mouthCoef=float(rng.uniform(2000, 6000)), noseCoef=float(rng.uniform(2000, 6000)),
              noseRadius=[0.0] + [float(x) for x in rng.uniform(0.5, 2.5, 5)], throatCutoff=float(rng.uniform(500, 3000)),
              throatVol=float(rng.uniform(0, 24)), usesModulation=int(rng.integers(0, 2)), mixOffset=float(rng.uniform(30, 60)))
    pd["tnMax"] = pd["tnMin"] + float(rng.uniform(5, 20))
    if broad:
        pd.update(controlRate=float(rng.choice([50.0, 125.0, 250.0, 333.0, 800.0, 1500.0])), tp=float(rng.uniform(5, 60)),
                  tnMin=float(rng.uniform(2, 20)), breathiness=float(rng.uniform(0, 40)), temperature=float(rng.uniform(10, 45)),
                  lossFactor=float(rng.uniform(0.0, 5.0)), apScale=float(rng.uniform(0.8, 8.0)), mouthCoef=float(rng.uniform(500, 9000)),
                  noseCoef=float(rng.uniform(500, 9000)), throatCutoff=float(rng.uniform(100, 8000)), throatVol=float(rng.uniform(0, 48)),
                  mixOffset=float(rng.uniform(10, 60)), noseRadius=[0.0] + [float(x) for x in rng.uniform(0.1, 3.0, 5)])
        pd["tnMax"] = pd["tnMin"] + float(rng.uniform(1, 35))
    voices = []
    for _ in range(int(rng.integers(1, 24))):
        n = int(rng.integers(0, maxframes)); knots = max(2, n // 8); t = np.linspace(0, knots - 1, max(n, 1))
        def track(lo, hi): return np.interp(t, np.arange(knots), rng.uniform(lo, hi, knots))
        fr = np.stack([track(-10, 6), track(0, 60), track(0, 20), track(0, 40), track(0, 7), track(500, 5000), track(200, 2500)] + [track(0.05, 2.5) for _ in range(8)] + [track(0.0, 1.2)], axis=1)
        if broad:
            fr = np.stack([track(-24, 24), track(0, 70), track(0, 45), track(0, 70), track(0, 7.999), track(100, 12000), track(50, 6000)] + [track(0.01, 3.0) for _ in range(8)] + [track(0.0, 2.0)], axis=1)
        voices.append(fr[:n])
    op = O.InputParams.from_dict(pd)
    if os.environ.get("FUZZ_VERBOSE"): print("seed %d: %d voices, rate %.0f, control rate %.0f, length %.2f" % (seed, len(voices), pd["outputRate"], pd["controlRate"], pd["length"]), flush=True)
    try:
        ref = [O.synthesize(op, np.asarray(v, np.float32).astype(np.float64)) for v in voices]
    except RuntimeError as e:            # the restatement refuses the parameters (e.g. tp + tnMax > 100 %): so must the library
        try:
            g.TRMBatch(g.TRMInputParameters.from_dict(pd))
            print("seed %d: oracle refuses (%s), library accepts" % (seed, e)); bad += 1
        except g.TrmError:
            pass
        continue
    if os.environ.get("FUZZ_ORACLE_ONLY"): continue
    for form in ("wide", "quad", "oct", "split"):
        try:
            b = g.TRMBatch(g.TRMInputParameters.from_dict(pd))
            if form == "split":
                # time-split launch (round 4): segments of a random length, warm-up by the library's rule; a tube that never
                # forgets (lossFactor ~ 0 in the broad set) is refused by name -- expected, not a finding
                b.set_time_split(int(rng.integers(3, 40)))
            else:
                b.set_kernel(form)
            pcm, ns, mx = b.synthesize(voices)
        except g.TrmError as e:
            if form == "split" and e.code == 10: continue           # TRM_ERANGE
            print("seed %d %s: %s" % (seed, form, e)); bad += 1; continue
        except Exception as e:
            print("seed %d %s: %s" % (seed, form, e)); bad += 1; continue
        for v, o in enumerate(ref):
            if int(ns[v]) != o["numberSamples"]:
                print("seed %d %s voice %d: count %d vs %d" % (seed, form, v, ns[v], o["numberSamples"])); bad += 1; continue
            if o["numberSamples"] == 0 or o["maximumSampleValue"] == 0: continue
            if cases.bandpass_unstable(np.asarray(voices[v], np.float32), o["derived"]["sampleRate"]): unstable += 1; continue   # outside the band-pass's domain
            r, a = cases.parity_error(pcm[v], o["samples"], o["maximumSampleValue"])
            if not r <= 1e-5 and a <= cases.ABS_FLOOR: floor += 1; continue      # a nearly silent voice, matched absolutely
            worst = max(worst, r)
            allr.append((r, seed, form, v, len(voices[v]), int(ns[v]), pd["outputRate"]))
            if r > worstBy.get(form, (0.0, -1, -1))[0]: worstBy[form] = (r, seed, v)
            if not r <= 1e-5:
                print("seed %d %s voice %d (%d frames, length %.1f, rate %.0f/%.0f): rms %.3e" % (seed, form, v, len(voices[v]), pd["length"], pd["outputRate"], pd["controlRate"], r)); bad += 1
print("done: seeds %d..%d, %d findings; worst normalised RMS %.3e; %d voice-runs outside the band-pass's domain (BW >= SR/2), %d nearly silent ones matched on the absolute floor %.0e"
      % (first, last, bad, worst, unstable, floor, cases.ABS_FLOOR))
print("worst per form: " + "; ".join("%s %.3e (seed %d voice %d)" % ((f,) + worstBy[f]) for f in sorted(worstBy)))
allr.sort(reverse=True)
print("the ten largest: " + "; ".join("%.2e (seed %d %s voice %d: %d frames, %d samples, %.0f Hz)" % t for t in allr[:10]))
if allr:
    rs = np.array([t[0] for t in allr])
    print("distribution over %d voice-runs: median %.2e, 99th percentile %.2e, above 5e-6: %d" % (len(rs), np.median(rs), np.percentile(rs, 99), int((rs > 5e-6).sum())))
