#!/usr/bin/env python3
"""A short utterance (26 frames = 0.1 s, 40 voices) launched directly vs replayed from a captured HIP graph
(trm_batch_set_timing(0) + torch.cuda.graph): host-side cost per launch and time to completion."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, cases, gnuspeech_amd as g
b = g.TRMBatch(g.TRMInputParameters.from_dict(cases.monet_default_params(44100.0)))
st = b.prepare_device(cases.config3_frames(40, nframes=26))
b.synthesize_device(st); torch.cuda.synchronize()
def timed(fn, n=300):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn(); torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(n): fn()
    t2 = time.perf_counter(); torch.cuda.synchronize(); t3 = time.perf_counter()
    return (t1 - t0) / n * 1e6, (t2 - t1) / n * 1e6, (t3 - t1) / n * 1e6
d = timed(lambda: b.synthesize_device(st))
b.set_timing(False)
d2 = timed(lambda: b.synthesize_device(st))
side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side): b.synthesize_device(st, side)
torch.cuda.current_stream().wait_stream(side)
gr = torch.cuda.CUDAGraph()
with torch.cuda.graph(gr): b.synthesize_device(st)
r = timed(gr.replay)
print("40 voices x 0.1 s, per launch in us (launch+sync | host cost of an async launch | back-to-back throughput):")
print("  direct, timing on : %.0f | %.0f | %.0f" % d)
print("  direct, timing off: %.0f | %.0f | %.0f" % d2)
print("  graph replay      : %.0f | %.0f | %.0f" % r)
