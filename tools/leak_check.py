#!/usr/bin/env python3
"""Soak: 1500 rounds of create / synthesize / destroy over batch, stream, multi-device and single-tube objects; device memory
in use and host RSS must not grow (round 1: 859.8 -> 859.8 MB, 1760.8 -> 1760.8 MB)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, cases, gnuspeech_amd as g, resource
rows = cases.load_gnuspeech_rows()
def mem(): 
    torch.cuda.synchronize(); free, total = torch.cuda.mem_get_info(); return (total - free) / 1e6
def rss(): return resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1e3
ips = [g.TRMInputParameters.from_dict(cases.monet_default_params(r)) for r in (44100.0, 16000.0)]
for ip in ips:                                   # warm the process-wide tables
    g.TRMBatch(ip).synthesize([rows[:50]]); s = g.TRMStream(ip, 2); s.push(np.stack([rows[:20], rows[:20]]).astype(np.float32)); s.finish()
    g.TRMMultiBatch(ip, [0, 0]).synthesize([rows[:30], rows[:40]])
m0, r0 = mem(), rss()
for it in range(1500):
    ip = ips[it & 1]
    b = g.TRMBatch(ip); b.synthesize([rows[: 20 + it % 60], rows[5:40]]); del b
    if it % 3 == 0:
        s = g.TRMStream(ip, 2); s.push(np.stack([rows[:12], rows[:12]]).astype(np.float32)); s.finish(); del s
    if it % 5 == 0:
        m = g.TRMMultiBatch(ip, [0, 0]); m.synthesize([rows[:30], rows[:40], rows[:9]]); del m
    if it % 7 == 0:
        dl = g.TRMDataList(); dl.inputParameters = ip; dl.values = [g.TRMParameters(r) for r in rows[:30]]
        t = g.TRMTubeModel.initWithInputData(dl); t.synthesize(); del t
m1, r1 = mem(), rss()
print("device memory in use: %.1f MB -> %.1f MB; host max RSS %.1f MB -> %.1f MB after 1500 create/synthesize/destroy rounds" % (m0, m1, r0, r1))
