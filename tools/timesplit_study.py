#!/usr/bin/env python3
"""How much warm-up does a time-split tube need?  (CPU; test infrastructure: tests/_emul + the oracle)

Runs the host model of the segmented path (tests/_emul: trm_emul_synthesize_split -- every segment after the first starts
from REST `warm` control periods early; oscillator position and noise index are exact) against the oracle for a set of
voices and warm-up lengths and prints the normalised RMS over the whole utterance and the worst normalised error of a
single sample.  usage: timesplit_study.py [seg_periods]
       timesplit_study.py --random N     N random / adversarial voices (closed mouth + velum, narrow constrictions, loud-then-quiet
                                         tracks) at Monet's defaults: the worst difference of the split path from the WHOLE-utterance
                                         path of the same arithmetic, by warm-up length (what the library's 1e-5 rule was chosen on)"""
import ctypes as C
import math
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import cases          # noqa: E402
import oracle_lib as O  # noqa: E402

E = C.CDLL(os.path.join(ROOT, "tests", "_emul", "libtrm_emul.so"))
E.trm_emul_synthesize_split.argtypes = [C.POINTER(O.InputParams), C.POINTER(C.c_float), C.c_size_t, C.POINTER(C.c_float), C.c_size_t,
                                        C.POINTER(C.c_uint32), C.POINTER(C.c_float), C.c_uint32, C.c_uint32]


def split(p, fr, seg, warm):
    fr = np.ascontiguousarray(fr, dtype=np.float32)
    cap = len(fr) * 800 + 2000
    out = np.zeros(cap, dtype=np.float32)
    n, m = C.c_uint32(), C.c_float()
    rc = E.trm_emul_synthesize_split(C.byref(p), fr.ctypes.data_as(C.POINTER(C.c_float)), len(fr), out.ctypes.data_as(C.POINTER(C.c_float)),
                                     cap, C.byref(n), C.byref(m), seg, warm)
    assert rc == 0, rc
    return out[:n.value]


def voices():
    rows = cases.load_gnuspeech_rows()
    mon = cases.monet_default_params(44100.0)
    out = []
    out.append(("gnuspeech.input x2 (686 fr)", mon, np.concatenate([rows, rows])))
    c4 = cases.config4_frames(6, lo=400, hi=700)
    for i, v in enumerate(c4[:3]):
        out.append(("config4 voice %d (%d fr)" % (i, len(v)), mon, v))
    c2 = cases.config2_frames(4, nframes=501)
    for i in range(3):
        out.append(("config2 static vowel %d" % i, mon, c2[i]))
    mv = [-12.0, 60.0, 0.0, 0.0, 5.5, 2500.0, 500.0, 0.8, 0.89, 0.99, 0.81, 0.76, 1.05, 1.23, 0.01, 0.1]
    out.append(("monet vowel, mouth closed (r8 .01), velum .1", mon, cases.static_frames(mv, 501)))
    mv2 = list(mv); mv2[15] = 0.0
    out.append(("mouth closed AND velum closed", mon, cases.static_frames(mv2, 501)))
    tr = cases.tract_default_params()
    out.append(("TRAcT vowel (loss 0.8, 100 Hz control)", tr, cases.static_frames(cases.TRACT_VOWEL_FRAME, 201)))
    fr = cases.static_frames([-12.0, 54.0, 6.0, 50.0, 5.4, 2500.0, 250.0, 0.8, 0.89, 0.99, 0.81, 0.76, 0.3, 1.23, 0.9, 0.1], 501)
    out.append(("fricative, BW 250 Hz", mon, fr))
    fr2 = fr.copy(); fr2[:, 6] = 60.0
    out.append(("fricative, BW 60 Hz", mon, fr2))
    lo = dict(mon); lo["lossFactor"] = 0.1
    out.append(("loss 0.1 %, gnuspeech.input", lo, np.concatenate([rows, rows])))
    return out


def random_study(N):
    rng = np.random.default_rng(1)
    pd = cases.monet_default_params(44100.0)
    p = O.InputParams.from_dict(pd)
    warms = (24, 27, 30, 33, 36)
    worst = {w: (0.0, None) for w in warms}
    rms_w = {w: 0.0 for w in warms}
    used = 0
    for it in range(N):
        n = 260
        knots = max(2, n // int(rng.integers(4, 40)))
        t = np.linspace(0, knots - 1, n)

        def track(lo, hi):
            return np.interp(t, np.arange(knots), rng.uniform(lo, hi, knots))
        kind = it % 4
        fr = np.stack([track(-10, 6), track(30, 60), track(0, 20), track(0, 40), track(0, 7), track(500, 5000), track(250, 2500)] +
                      [track(0.05, 2.5) for _ in range(8)] + [track(0.0, 1.2)], axis=1)
        if kind == 1:       # mouth closed, velum (nearly) closed
            fr[:, 14] = 0.01
            fr[:, 15] = rng.choice([0.0, 0.02, 0.1])
        if kind == 2:       # static, one narrow constriction
            fr[:] = fr[0]
            fr[:, 7 + int(rng.integers(0, 8))] = 0.02
        if kind == 3:       # loud, then quiet: what the warm-up forgets is large against what follows
            fr[:, 1] = np.concatenate([np.full(120, 60.0), np.linspace(60, 20, 140)])
        whole = split(p, fr, 1 << 30, 0).astype(np.float64)
        mx = np.abs(whole).max()
        if mx < 1e-3:       # nearly silent voices: fp32 noise (the parity tools match them on an absolute floor)
            continue
        used += 1
        for w in warms:
            y = split(p, fr, 40, w).astype(np.float64)
            d = np.abs(y - whole).max() / mx
            rms_w[w] = max(rms_w[w], math.sqrt(np.mean(((y - whole) / mx) ** 2)))
            if d > worst[w][0]:
                worst[w] = (d, (it, kind))
    print("%d voices (%d not nearly silent), segments of 40 control periods, split vs whole utterance of the same fp32 arithmetic:" % (N, used))
    for w in warms:
        print("  warm-up %d periods: worst single sample %.2e of the maximum (voice %s), worst RMS %.2e" % (w, worst[w][0], worst[w][1], rms_w[w]))


def main():
    if len(sys.argv) > 2 and sys.argv[1] == "--random":
        return random_study(int(sys.argv[2]))
    seg = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    print("segments of %d control periods; columns: warm-up in control periods -> nRMS (worst single sample / max)" % seg)
    for name, pd, fr in voices():
        p = O.InputParams.from_dict(pd)
        f32 = np.asarray(fr, dtype=np.float32)
        o = O.synthesize(p, f32.astype(np.float64))
        mx = o["maximumSampleValue"]
        damping = 1.0 - pd["lossFactor"] / 100.0
        cp = int(o["derived"]["controlPeriod"])
        line = []
        base = split(p, f32, 1 << 30, 0)
        e0 = (base.astype(np.float64) - o["samples"]) / mx
        line.append("unsplit %.2e" % math.sqrt(np.mean(e0 * e0)))
        for warm in (5, 10, 20, 30, 40, 60):
            y = split(p, f32, seg, warm)
            e = (y.astype(np.float64) - o["samples"]) / mx
            line.append("%d: %.2e (%.1e)" % (warm, math.sqrt(np.mean(e * e)), np.abs(e - e0).max()))
        print("%-46s damping %.4f cp %s | %s" % (name, damping, cp, "  ".join(line)))


if __name__ == "__main__":
    main()
