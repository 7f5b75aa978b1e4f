#!/usr/bin/env python3
"""Randomized check of the control-track generator (trm_tracks_kernel): GPU frames == oracle frames bit
for bit over random event lists (regular and irregular event times, absent targets, special-event offsets, all intonation
switches, time ranges), many lists per launch.   usage: fuzz_events.py first_seed last_seed"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import cases
import oracle_lib as O
import gnuspeech_amd as g
from test_events import random_events, settings, _to_g

first, last = int(sys.argv[1]), int(sys.argv[2])
b = g.TRMBatch(g.TRMInputParameters.from_dict(cases.monet_default_params()))
bad = 0
for seed in range(first, last):
    rng = np.random.default_rng(70000 + seed)
    smooth = int(rng.integers(0, 2))
    start, end = (0, 0) if rng.random() < 0.6 else (int(rng.integers(0, 60)), int(rng.integers(60, 400)))
    s = settings(int(rng.integers(0, 2)), int(rng.integers(0, 2)), smooth, int(rng.integers(0, 2)), dev=float(rng.uniform(0, 2)),
                 cutoff=float(rng.uniform(0.5, 8)), pitch=float(rng.uniform(-15, 0)), start=start, end=end)
    lists = []
    for _ in range(int(rng.integers(1, 40))):
        n = int(rng.integers(0, 45))
        if n < 1:
            lists.append((np.zeros(0, np.uint32), np.zeros((0, 36))))
            continue
        times, vals = random_events(rng, max(n, 1), span=int(rng.integers(8, 60)), nan_frac=float(rng.uniform(0.1, 0.9)), smooth=bool(smooth))
        if rng.random() < 0.4 and n > 1:        # irregular times, several events per frame interval, equal times
            times = np.concatenate([[0], np.cumsum(rng.integers(0, 11, size=n - 1))]).astype(np.uint32)
        lists.append((times[:n], vals[:n]))
    st = b.prepare_events_device(lists, _to_g(g, s))
    b.generate_frames_device(st)
    torch.cuda.synchronize()
    frames = st["frames"].cpu().numpy()
    got_n = st["nframes_generated"].cpu().numpy()
    foff = st["frame_offset_host"] if "frame_offset_host" in st else st["frame_offset"].cpu().numpy()
    for v, (times, vals) in enumerate(lists):
        want = O.generate_frames(times, vals, s) if len(times) else np.zeros((0, 16), np.float32)
        if int(got_n[v]) != want.shape[0]:
            print("seed %d list %d: %d frames vs %d" % (seed, v, got_n[v], want.shape[0])); bad += 1; continue
        got = frames[int(foff[v]):int(foff[v]) + want.shape[0]]
        same = (got.view(np.uint32) == want.view(np.uint32)) | (np.isnan(got) & np.isnan(want))     # (NaN payloads differ between x86 and the GPU)
        if not same.all():
            d = ~same
            both_nan = np.isnan(got[d]) & np.isnan(want[d])
            zeros = (got[d] == 0) & (want[d] == 0)
            print("seed %d list %d (%d events): %d values differ (%d NaN payloads, %d signed zeros), columns %s, max abs %.3e" % (
                seed, v, len(times), int(d.sum()), int(both_nan.sum()), int(zeros.sum()), sorted(set(np.nonzero(d)[1].tolist())), np.nanmax(np.abs(got - want)))); bad += 1
print("done: seeds %d..%d, %d findings" % (first, last, bad))
