import os, sys, time
sys.path.insert(0, "."); sys.path.insert(0, "tests"); sys.path.insert(0, "tools")
import numpy as np, torch, cases, gnuspeech_amd as g
pd = cases.monet_default_params(44100.0)
for V in (20480, 24576, 40960, 49152):
    fr = cases.config2_frames(V, nframes=251)
    b = g.TRMBatch(g.TRMInputParameters.from_dict(pd))
    st = b.prepare_device(fr)
    for sp in ("off", 110, 74, 55):
        b.set_time_split(sp)
        b.synthesize_device(st); torch.cuda.synchronize(); b.kernel_time_ms()
        for _ in range(4): b.synthesize_device(st)
        torch.cuda.synchronize()
        km, n = b.kernel_time_ms()
        print("%6d voices split %-4s -> %s %.3f ms  %.3e samples/s" % (V, sp, b.last_time_split, km / n, st["total_out"] / (km / n * 1e-3)), flush=True)
    del st, b; torch.cuda.empty_cache()
