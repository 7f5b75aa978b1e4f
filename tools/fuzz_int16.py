#!/usr/bin/env python3
"""Randomized check of the output normalisation (trm_int16_kernel vs the oracle's restatement of TRMTubeModel.m:370-389 /
:515-540): random volume, balance, channels, both scalings, samples incl. values whose scaled result leaves the int16 range
(the reference's cast wraps).   usage: fuzz_int16.py first_seed last_seed"""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import cases
import oracle_lib as O
import gnuspeech_amd as g
from gnuspeech_amd._capi import lib, check

first, last = int(sys.argv[1]), int(sys.argv[2])
bad = 0
for seed in range(first, last):
    rng = np.random.default_rng(30000 + seed)
    pd = cases.monet_default_params(44100.0)
    pd.update(channels=int(rng.integers(1, 3)), balance=float(rng.uniform(-1, 1)), volume=float(rng.uniform(0, 60)))
    ip = g.TRMInputParameters.from_dict(pd)
    b = g.TRMBatch(ip)
    op = O.InputParams.from_dict(pd)
    V = int(rng.integers(1, 9))
    ns = rng.integers(0, 3000, V).astype(np.uint32)
    off = np.concatenate([[0], np.cumsum(ns[:-1])]).astype(np.uint64)
    total = int(ns.sum()) + 1
    pcm = (rng.standard_normal(total) * rng.choice([1e-4, 1e-2, 1.0, 30.0])).astype(np.float32)
    mx = np.array([max(1e-9, float(np.abs(pcm[int(o):int(o) + int(n)]).max()) * float(rng.choice([1.0, 1.0, 0.5, 2.0]))) if n else 1.0
                   for o, n in zip(off, ns)], dtype=np.float32)
    ch = 2 if pd["channels"] == 2 else 1
    dev = "cuda"
    d_pcm, d_off = torch.from_numpy(pcm).to(dev), torch.from_numpy(off.astype(np.int64)).to(dev)
    d_ns, d_mx = torch.from_numpy(ns.astype(np.int32)).to(dev), torch.from_numpy(mx).to(dev)
    for wav in (0, 1):
        d16 = torch.zeros(total * ch, dtype=torch.int16, device=dev)
        check(lib().trm_batch_scale_to_int16_device(b._h, V, d_pcm.data_ptr(), d_off.data_ptr(), d_ns.data_ptr(), d_mx.data_ptr(),
                                                    d16.data_ptr(), wav, C.c_void_p(torch.cuda.current_stream().cuda_stream)))
        torch.cuda.synchronize()
        got = d16.cpu().numpy()
        for v in range(V):
            n, o = int(ns[v]), int(off[v])
            if n == 0:
                continue
            ref = np.asarray(O.scale_int16(op, pcm[o:o + n].astype(np.float64), float(mx[v]), for_wav_data=bool(wav)))
            if not np.array_equal(got[o * ch:(o + n) * ch], ref):
                k = np.nonzero(got[o * ch:(o + n) * ch] != ref)[0]
                print("seed %d voice %d wav %d ch %d: %d of %d differ, first at %d: %d vs %d" % (seed, v, wav, ch, len(k), n * ch, k[0], got[o * ch + k[0]], ref[k[0]])); bad += 1
print("done: seeds %d..%d, %d findings" % (first, last, bad))
