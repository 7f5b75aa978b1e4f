#!/usr/bin/env python3
"""Where does a pipeline step spend its cycles?  Loads the TRM_STAMP diagnostic build
(gnuspeech_amd/libtrm_hip_stamp.so: `make -C gnuspeech_amd/csrc stamp`) and prints, per pipeline
role, the cycles spent working vs waiting at the step barrier (median over workgroups).
Read the SHARES, not the run time: stamps perturb the kernel."""
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

V = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 0.25
wl = sys.argv[3] if len(sys.argv) > 3 else "static"
form = sys.argv[4] if len(sys.argv) > 4 else "auto"
nframes = int(round(secs * 250)) + 1
fr = cases.config2_frames(V, nframes=nframes) if wl == "static" else cases.config3_frames(V, nframes=nframes)
b = g.TRMBatch(g.TRMInputParameters.from_dict(cases.monet_default_params(44100.0)))
b.set_kernel(form)
st = b.prepare_device(fr)
for _ in range(2):
    b.synthesize_device(st)
torch.cuda.synchronize()
L = g.lib()
quad = b.last_kernel in ("quad", "oct")
octInst = b.last_kernel == "oct"     # eight lanes per voice, 8 voices per workgroup
nwg = (V + 7) // 8 if octInst else (V + 15) // 16 if quad else (V + 63) // 64
NR = 6 if quad else 7
buf = np.zeros(nwg * NR * 8, dtype=np.uint64)
L.trm_debug_stamps.argtypes = [C.c_void_p, C.c_size_t]
assert L.trm_debug_stamps(buf.ctypes.data, buf.size) == 0
s = buf.reshape(nwg, NR, 8).astype(np.float64)
ntube = (nframes - 1) * b.derived["controlPeriod"] + 26
names = ["osc", "mix", "coef0", "coef1", "tube", "convert"] if quad else ["osc", "mix", "coef0", "coef1", "tube", "convert0", "convert1"]
print("kernel form %s; voices %d, %d tube samples; cycles per tube sample (median over %d workgroups)" % (b.last_kernel, V, ntube, nwg))
for r in range(NR):
    w, q = np.median(s[:, r, 0]) / ntube, np.median(s[:, r, 1]) / ntube
    nl, ex, mxs = np.median(s[:, r, 5]), np.median(s[:, r, 6]), np.median(s[:, r, 7])
    print("  %-8s work %7.0f  barrier-wait %7.0f  total %7.0f   long steps %5.0f (excess %6.0f cycles each), longest %6.0f" % (names[r], w, q, w + q, nl, ex / max(nl, 1), mxs))
# HW_ID (gfx9 layout): wave slot [3:0], SIMD [5:4], CU [11:8]: which roles share a SIMD
# (one-voice-per-lane kernel: the convert waves' slots hold their sub-phase times instead, so only the first five roles are listed)
hw = buf.reshape(nwg, NR, 8)[:, :(NR if quad else 5), 2].astype(np.int64)
simd = (hw >> 4) & 3
pat = {}
for row in simd:
    pat[tuple(row)] = pat.get(tuple(row), 0) + 1
print("  SIMD of (%s): %s" % (", ".join(names[:simd.shape[1]]), "; ".join("%s x%d" % ("".join(map(str, k)), n) for k, n in sorted(pat.items(), key=lambda kv: -kv[1])[:8])))
if quad:
    # which workgroups shared a CU (HW_ID: CU [11:8], SH [12], SE [15:13]; the XCD is not in HW_ID, so up to 8 CUs
    # alias one key: read the PATTERNS of the tube wave's SIMD among the workgroups of a key, not the counts)
    cu = (hw[:, 4] >> 8) & 0xFF
    tube_simd = simd[:, 4]
    print("  tube wave's SIMD, histogram over workgroups: %s" % np.bincount(tube_simd, minlength=4).tolist())
    # when did each workgroup's tube wave run (100 MHz clock)?  co-resident workgroups start together
    born = buf.reshape(nwg, NR, 8)[:, 4, 3].astype(np.float64); died = buf.reshape(nwg, NR, 8)[:, 4, 4].astype(np.float64)
    t0 = born.min()
    print("  tube wave start (us after the first): quartiles %s; run time (us): median %.0f; whole launch %.0f us" % (
        np.percentile((born - t0) / 100.0, [0, 25, 50, 75, 100]).round(0).tolist(), np.median(died - born) / 100.0, (died.max() - t0) / 100.0))
    hist, edges = np.histogram((born - t0) / 100.0, bins=12)
    print("  start-time histogram (us): " + ", ".join("%.0f:%d" % (edges[i], hist[i]) for i in range(len(hist)) if hist[i]))
    rt = (died - born) / 100.0
    print("  run time by start group: " + ", ".join("start<%.0f: n=%d median run %.0f us" % (hi, int(((born - t0) / 100.0 < hi).sum()), float(np.median(rt[(born - t0) / 100.0 < hi]))) for hi in (10.0,)) +
          "; later starters: n=%d median run %.0f us" % (int(((born - t0) / 100.0 >= 10.0).sum()), float(np.median(rt[(born - t0) / 100.0 >= 10.0]))))
if not quad:
    sub = np.median(s[:, 5, 2:], axis=0) / ntube
    print("  convert0 sub-phases (cycles per tube sample): reads-issue %.0f, readlanes %.0f, dot %.0f, stores %.0f, tile+block-end %.0f" % tuple(sub[:5]))
