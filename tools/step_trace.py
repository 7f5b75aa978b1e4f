#!/usr/bin/env python3
"""Per-step work of every pipeline role of workgroup 0 (stamp build with -DTRM_STAMP_TRACE): how much of the step is the
slowest role of THAT step (what one barrier for all waves pays) against the slowest role on average (what a decoupled
pipeline would pay).   usage: TRM_STAMP_LIB=.../libtrm_stamp_trace.so step_trace.py [voices] [seconds]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["TRM_LIB"] = os.environ.get("TRM_STAMP_LIB") or os.path.join(ROOT, "gnuspeech_amd", "libtrm_stamp_trace.so")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, cases
import gnuspeech_amd as g
V = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
nframes = int(round(secs * 250)) + 1
b = g.TRMBatch(g.TRMInputParameters.from_dict(cases.monet_default_params(44100.0)))
b.set_kernel("quad")
st = b.prepare_device(cases.config2_frames(V, nframes=nframes))
for _ in range(2):
    b.synthesize_device(st)
torch.cuda.synchronize()
L = g.lib()
buf = np.zeros(400000 + 7 * 4096, dtype=np.uint64)
L.trm_debug_stamps.argtypes = [C.c_void_p, C.c_size_t]
assert L.trm_debug_stamps(buf.ctypes.data, buf.size) == 0
tr = buf[400000:400000 + 6 * 4096].reshape(6, 4096).astype(np.float64)
n = int((tr[4] > 0).sum())
tr = tr[:, 8:n - 8]                     # (steady state: without the pipeline's fill and drain)
names = ["osc", "mix", "area", "fric", "tube", "convert"]
print("workgroup 0, %d steady steps; work cycles per step: " % tr.shape[1] + ", ".join("%s %.0f +- %.0f" % (names[r], tr[r].mean(), tr[r].std()) for r in range(6)))
mx = tr.max(axis=0)
print("mean over steps of the slowest role of the step: %.0f; slowest role on average: %.0f (%s): one barrier for all waves pays %.1f %% for the jitter"
      % (mx.mean(), tr.mean(axis=1).max(), names[int(tr.mean(axis=1).argmax())], 100.0 * (mx.mean() / tr.mean(axis=1).max() - 1.0)))
print("which role is the slowest of a step: " + ", ".join("%s %.0f %%" % (names[r], 100.0 * (tr.argmax(axis=0) == r).mean()) for r in range(6)))
