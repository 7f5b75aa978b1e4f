#!/usr/bin/env python3
"""What does the per-step barrier cost?  Per-step work cycles of every role of workgroup 0 (a -DTRM_STAMP -DTRM_STAMP_TRACE build of the
kernels: TRM_STAMP_LIB=...), and three totals over the traced steps:
  lockstep   sum over steps of the slowest role's work        (what one barrier per step costs at least)
  elastic-1  every role may run ONE step ahead of the slowest (it starts step s when all have finished s-2): needs every hand-off
             one step later and two slots deeper than the lockstep pipeline has them -- an upper bound on what hand-shakes could buy
  free       the busiest role's total                          (unbounded buffering)
usage: step_trace.py <voices> <seconds> <static|timevarying> <wide|quad|oct>"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["TRM_LIB"] = os.environ["TRM_STAMP_LIB"]
import numpy as np, torch
import cases
import gnuspeech_amd as g
V, sec, kind, form = int(sys.argv[1]), float(sys.argv[2]), sys.argv[3], sys.argv[4]
nframes = int(round(sec * 250)) + 1
fr = cases.config2_frames(V, nframes=nframes) if kind == "static" else cases.config3_frames(V, nframes=nframes)
b = g.TRMBatch(g.TRMInputParameters.from_dict(cases.monet_default_params(44100.0)))
b.set_kernel(form)
st = b.prepare_device(fr)
for _ in range(2):
    b.synthesize_device(st)
torch.cuda.synchronize()
L = g.lib()
buf = np.zeros(400000 + 8 * 4096, dtype=np.uint64)
L.trm_debug_stamps.argtypes = [C.c_void_p, C.c_size_t]
assert L.trm_debug_stamps(buf.ctypes.data, buf.size) == 0
NR = 7 if form == "wide" else 6
w = buf[400000:400000 + NR * 4096].reshape(NR, 4096).astype(np.float64)
n = int(min((w[r] > 0).sum() for r in range(NR)))
w = w[:, 8:n - 8]
S = w.shape[1]
print("%s, %d voices: %d traced steps of workgroup 0; mean work per step by role: %s" % (form, V, S, np.round(w.mean(axis=1)).astype(int).tolist()))
lock = w.max(axis=0).sum()
free = w.sum(axis=1).max()
# elastic: role r may start step s when it finished s-1 and every other role finished step s-2 (double buffers, lags of one step folded in)
t = np.zeros((NR, S + 2))
for s in range(S):
    gate = t[:, s].max() if s >= 1 else 0.0      # everyone finished step s-2 (index s holds the finish time of step s-2 ... s offset 2)
    for r in range(NR):
        t[r, s + 2] = max(t[r, s + 1], gate) + w[r, s]
el = t[:, S + 1].max()
print("lockstep %.0f cycles per step, elastic-1 %.0f (%.1f %% less), free %.0f (%.1f %% less)" % (lock / S, el / S, 100 * (1 - el / lock), free / S, 100 * (1 - free / lock)))

# who is the slowest, and when: share of steps each role is the maximum in, its mean excess over the busiest role's mean there,
# and the work of each role by position in the control period (steps per period = control period / samples per step)
names = ["osc", "mix", "coef0", "coef1", "tube", "convert0", "convert1"][:NR] if form == "wide" else ["osc", "mix", "area", "fric", "tube", "convert"]
am = w.argmax(axis=0)
busiest = w.mean(axis=1).max()
for r in range(NR):
    sel = am == r
    if sel.any():
        print("  %-8s slowest in %4.1f %% of the steps, there %5.0f cycles (its mean %5.0f, std %4.0f)" % (names[r], 100 * sel.mean(), w[r, sel].mean(), w[r].mean(), w[r].std()))
spp = float(b.derived["controlPeriod"]) / {"wide": 2, "quad": 4, "oct": 8}[form]
if abs(spp - round(spp)) < 1e-9 and spp >= 2:
    P = int(round(spp))
    ph = (np.arange(S) + 8) % P
    print("  mean of the step's maximum by step within the control period (%d steps): %s" % (P, [int(w.max(axis=0)[ph == q].mean()) for q in range(P)]))
    for r in range(NR):
        print("    %-8s %s" % (names[r], [int(w[r, ph == q].mean()) for q in range(P)]))
if "--save" in sys.argv:
    np.save(sys.argv[sys.argv.index("--save") + 1], w)
if "-v" in sys.argv:
    for s0 in range(200, 264):
        print("   step %4d  %s   max %s" % (s0 + 8, " ".join("%5d" % int(w[r, s0]) for r in range(NR)), names[int(w[:, s0].argmax())]))
