#!/usr/bin/env python3
"""When does every workgroup of a time-split launch start and end?  (diagnostic build: make -C gnuspeech_amd/csrc stamp)
usage: split_birth_probe.py <segment periods> [voices]      -- configs[3]'s ragged batch, longest first"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["TRM_LIB"] = os.environ.get("TRM_STAMP_LIB") or os.path.join(ROOT, "gnuspeech_amd", "libtrm_hip_stamp.so")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import cases  # noqa: E402
import gnuspeech_amd as g  # noqa: E402

S = int(sys.argv[1])
V = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
frames = sorted(cases.config4_frames(V, seed=20250119), key=len, reverse=True)
b = g.TRMBatch(g.TRMInputParameters.from_dict(cases.monet_default_params(44100.0)))
b.set_kernel("wide")
b.set_kernel("auto")
b.set_time_split(S)
st = b.prepare_device(frames)
for _ in range(3):
    b.synthesize_device(st)
torch.cuda.synchronize()
sp, warm = b.last_time_split
ncol = (V + 63) // 64
P = max(len(u) for u in frames) - 1
nseg = 1 + max(0, -(-(P - (sp + warm)) // sp))
grid = ncol * nseg
L = g.lib()
buf = np.zeros(grid * 7 * 8, dtype=np.uint64)
L.trm_debug_stamps.argtypes = [C.c_void_p, C.c_size_t]
assert L.trm_debug_stamps(buf.ctypes.data, buf.size) == 0
s = buf.reshape(grid, 7, 8)
born, died = s[:, 4, 3].astype(np.float64) / 100.0, s[:, 4, 4].astype(np.float64) / 100.0     # tube wave, us
ok = born > 0
t0 = born[ok].min()
run = died - born
live = ok & (run > 50.0)
print("split (%d, %d), kernel %s: grid %d = %d columns x %d segments; %d workgroups ran more than 50 us" % (sp, warm, b.last_kernel, grid, ncol, nseg, int(live.sum())))
print("  start of the live ones (us after the first): quartiles %s" % np.percentile(born[live] - t0, [0, 25, 50, 75, 90, 100]).round(0).tolist())
print("  run time of the live ones (us): quartiles %s;   launch ends at %.0f us" % (np.percentile(run[live], [0, 25, 50, 75, 100]).round(0).tolist(), died[ok].max() - t0))
late = np.nonzero(live & (born - t0 > 100.0))[0]
print("  %d live workgroups started more than 100 us late: %s" % (len(late), ", ".join("wg %d (seg %d col %d) at %.0f ran %.0f" % (w, w // ncol, w % ncol, born[w] - t0, run[w]) for w in late[:12])))
hw = s[:, 4, 2].astype(np.int64)
cu = (hw >> 8) & 0xF
se = (hw >> 13) & 0x7
print("  empties: %d; their run time median %.1f us" % (int((ok & ~live).sum()), float(np.median(run[ok & ~live])) if (ok & ~live).any() else 0.0))
