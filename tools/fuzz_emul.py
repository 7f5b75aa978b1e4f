#!/usr/bin/env python3
"""CPU rehearsal of tools/fuzz_parity.py: the same randomized parameter sets and tracks, run through the HOST MODEL of
the kernels' arithmetic (tests/_emul: trm_lane.h / trm_quad.h compiled for the CPU, test infrastructure) instead of the
GPU, against the oracle.  Up-sampling parameter sets only (the host model has no down-sampling branch).
usage: fuzz_emul.py first last [maxframes] [broad]     prints every voice above 1e-5 and the worst seen"""
import ctypes as C, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import cases, oracle_lib as O

SRC = os.path.join(ROOT, "tests", "_emul", "trm_emul.cc")
LIB = os.path.join(ROOT, "tests", "_emul", "libtrm_emul.so")
csrc = os.path.join(ROOT, "gnuspeech_amd", "csrc")
deps = [SRC] + [os.path.join(csrc, f) for f in ("trm_lane.h", "trm_quad.h", "trm_setup.cc", "trm_setup.h")]
if not os.path.exists(LIB) or any(os.path.getmtime(d) > os.path.getmtime(LIB) for d in deps):
    subprocess.check_call(["g++", "-O2", "-ffp-contract=off", "-fPIC", "-shared", "-std=c++17", "-o", LIB, SRC, os.path.join(csrc, "trm_setup.cc"), "-lm"])
E = C.CDLL(LIB)
sig = [C.POINTER(O.InputParams), C.POINTER(C.c_float), C.c_size_t, C.POINTER(C.c_float), C.c_size_t, C.POINTER(C.c_uint32), C.POINTER(C.c_float), C.POINTER(C.c_float)]
E.trm_emul_synthesize.argtypes = sig
E.trm_emul_synthesize_quad.argtypes = sig

first, last = int(sys.argv[1]), int(sys.argv[2])
maxframes = int(sys.argv[3]) if len(sys.argv) > 3 else 300
broad = len(sys.argv) > 4 and sys.argv[4] == "broad"
bad = 0; worst = 0.0; nv = 0; unstable = 0; floor = 0
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
    for vi, fr in enumerate(voices):
        f32 = np.ascontiguousarray(fr, dtype=np.float32)
        if len(f32) < 2: continue
        try:
            o = O.synthesize(op, f32.astype(np.float64))
        except RuntimeError:
            break
        if o["numberSamples"] == 0 or o["maximumSampleValue"] == 0: continue
        if cases.bandpass_unstable(f32, o["derived"]["sampleRate"]): unstable += 1; continue      # outside the band-pass's domain
        for name, fn in (("lane", E.trm_emul_synthesize), ("quad", E.trm_emul_synthesize_quad)):
            cap = o["numberSamples"] + 64
            out = np.zeros(cap, dtype=np.float32); n, m = C.c_uint32(), C.c_float()
            rc = fn(C.byref(op), f32.ctypes.data_as(C.POINTER(C.c_float)), len(f32), out.ctypes.data_as(C.POINTER(C.c_float)), cap, C.byref(n), C.byref(m), None)
            if rc: break          # down-sampling parameters / control period too short for the four-lane form
            if n.value != o["numberSamples"]:
                print("seed %d voice %d %s: count %d vs %d" % (seed, vi, name, n.value, o["numberSamples"])); bad += 1; continue
            r, a = cases.parity_error(out[:n.value], o["samples"], o["maximumSampleValue"])
            nv += 1
            if not r <= 1e-5 and a <= cases.ABS_FLOOR: floor += 1; continue
            worst = max(worst, r)
            if not r <= 1e-5:
                print("seed %d voice %d %s (%d frames, length %.1f, rate %.0f/%.0f, apScale %.2f, loss %.2f): rms %.3e (max %.3e)" %
                      (seed, vi, name, len(f32), pd["length"], pd["outputRate"], pd["controlRate"], pd["apScale"], pd["lossFactor"], r, o["maximumSampleValue"]), flush=True)
                bad += 1
print("done: seeds %d..%d, %d voice-forms, worst %.3e, %d findings; %d voices outside the band-pass's domain, %d on the absolute floor" % (first, last, nv, worst, bad, unstable, floor))
