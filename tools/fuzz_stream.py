#!/usr/bin/env python3
"""Randomized check of the streaming entry (trm_stream_*): random parameters (both converter branches), random cuts.
A chunked stream must equal the single-push stream bit for bit, and the one-shot batch result to rounding (same count,
except where the reference's one-shot converter ends on its extra lap).
TRM_TUBE_KERNEL=quad|wide picks the streaming kernel form; `tract` as third argument streams in TRAcT's loop order
(TRM_STREAM_MODE_TRACT) and compares with the oracle in that order (trm_oracle_synthesize_tract) at 1e-5 instead of with the
one-shot batch.
usage: fuzz_stream.py first_seed last_seed [tract]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import cases
import gnuspeech_amd as g
from gnuspeech_amd import shard

def d_rate(ip):
    return shard.derive(ip)["sampleRate"]


first, last = int(sys.argv[1]), int(sys.argv[2])
tract = len(sys.argv) > 3 and sys.argv[3] == "tract"
mode = "tract" if tract else "framework"
if tract:
    import oracle_lib as O
bad = 0
rows = cases.load_gnuspeech_rows()
for seed in range(first, last):
    rng = np.random.default_rng(9000 + seed)
    pd = cases.monet_default_params(float(rng.choice([44100.0, 22050.0, 16000.0, 8000.0, 11025.0, 48000.0, 32000.0])))
    pd.update(length=float(rng.uniform(12.0, 25.0)), controlRate=float(rng.choice([100.0, 250.0, 500.0])), waveform=int(rng.integers(0, 2)),
              usesModulation=int(rng.integers(0, 2)))
    V, n = int(rng.integers(1, 20)), int(rng.integers(3, 260))
    start = rng.integers(0, len(rows), V)
    fr = np.stack([np.concatenate([rows, rows])[s:s + n] for s in start]).astype(np.float32)
    fr[:, :, 0] += rng.uniform(-4, 4, (V, 1)).astype(np.float32)
    ip = g.TRMInputParameters.from_dict(pd)
    try:
        s1 = g.TRMStream(ip, nvoices=V, mode=mode)
    except g.TrmError as e:
        continue                                            # (rate ratio above 4: streams refuse)
    def run(stream, cuts):
        parts, pos = [], 0
        for c in cuts:
            o, _m = stream.push(fr[:, pos:pos + c]); parts.append(o); pos += c
        o, _m = stream.finish(); parts.append(o)
        return np.concatenate(parts, axis=1)
    whole = run(s1, [n])
    cuts = []
    left = n
    while left > 0:
        c = int(min(left, rng.integers(1, max(2, n // 3 + 1)))); cuts.append(c); left -= c
    got = run(g.TRMStream(ip, nvoices=V, mode=mode), cuts)
    if got.shape != whole.shape or not np.array_equal(got.view(np.uint32), whole.view(np.uint32)):
        print("seed %d: chunked != whole (rate %.0f, length %.1f, V %d, n %d, cuts %s)" % (seed, pd["outputRate"], pd["length"], V, n, cuts)); bad += 1
        continue
    if tract:
        op = O.InputParams.from_dict(pd)
        for v in range(V):
            o = O.synthesize(op, np.concatenate([fr[v][:1], fr[v]]).astype(np.float64), tract=True)
            if whole.shape[1] != o["numberSamples"]:
                print("seed %d voice %d: count %d vs oracle %d" % (seed, v, whole.shape[1], o["numberSamples"])); bad += 1; continue
            if o["maximumSampleValue"] == 0: continue
            e = (whole[v].astype(np.float64) - o["samples"]) / o["maximumSampleValue"]
            r = float(np.sqrt(np.mean(e * e)))
            if not r <= 1e-5 and not cases.bandpass_unstable(fr[v], d_rate(ip)):
                print("seed %d voice %d: tract-order stream vs oracle rms %.3e (rate %.0f, length %.1f)" % (seed, v, r, pd["outputRate"], pd["length"])); bad += 1
        continue
    b = g.TRMBatch(ip); b.set_kernel("quad")
    pcm, ns, mx = b.synthesize(list(fr))
    d = shard.derive(ip)
    plain = (((n - 1) * d["controlPeriod"] + 2 * d["padSize"]) * 65536 + d["timeRegisterIncrement"] - 1) // d["timeRegisterIncrement"]
    if whole.shape[1] != plain:
        print("seed %d: stream count %d vs %d" % (seed, whole.shape[1], plain)); bad += 1; continue
    for v in range(V):
        if mx[v] == 0: continue
        m = min(plain, int(ns[v]))
        e = (whole[v, :m].astype(np.float64) - pcm[v][:m]) / float(mx[v])
        r = float(np.sqrt(np.mean(e * e)))
        if not r <= 4e-6:
            print("seed %d voice %d: stream vs one-shot rms %.3e (rate %.0f, length %.1f)" % (seed, v, r, pd["outputRate"], pd["length"])); bad += 1
print("done: seeds %d..%d, %d findings" % (first, last, bad))
