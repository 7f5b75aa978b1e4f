#!/usr/bin/env python3
"""Launch-to-launch determinism: the same resident batch synthesized N times must give the same bits every time (no role
of the pipeline may depend on timing: the hand-shakes between waves carry data, never values).  Every kernel form.
usage: determinism_soak.py [launches per form]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import cases
import gnuspeech_amd as g

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
pd = cases.monet_default_params(44100.0)
for form, voices in (("oct", 4096), ("quad", 8192), ("wide", 16384)):
    b = g.TRMBatch(g.TRMInputParameters.from_dict(pd)); b.set_kernel(form)
    st = b.prepare_device(cases.config3_frames(voices, nframes=126))
    b.synthesize_device(st); torch.cuda.synchronize()
    assert b.last_kernel == form
    ref, refmx = st["out"].clone(), st["max_sample"].clone()
    bad = 0
    for i in range(n):
        st["out"].zero_()
        b.synthesize_device(st); torch.cuda.synchronize()
        if not (torch.equal(st["out"], ref) and torch.equal(st["max_sample"], refmx)):
            bad += 1
    print("%s form, %d voices x 0.5 s: %d launches, %d differing from the first" % (form, voices, n, bad))
    assert bad == 0
