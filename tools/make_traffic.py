#!/usr/bin/env python3
"""Turns the PMC passes of tools/profile_round.sh into profiles/traffic_r02.json -- the file bench.py quotes
`roofline.traffic` and `roofline.valu_issue` from -- and STAMPS it with the hash of the kernel sources it was measured
on (bench.kernel_source_hash()): bench.py refuses a file whose stamp differs from the sources it is running.
HBM bytes per launch as MI355X_MICROARCH.md prescribes: WRITE_SIZE and FETCH_SIZE collected in SEPARATE passes, both in
KB, FETCH_SIZE doubled on gfx950 (it counts half of wide coalesced reads).
usage: make_traffic.py <summary.txt from rocprof_summary.py> <voices> <frames> <static|timevarying> <quad|wide> <issue cycles per VALU> <out.json>"""
import json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench

summary, voices, frames, kind, form, cyc, out = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], sys.argv[5], float(sys.argv[6]), sys.argv[7]
vals = {}
for ln in open(summary):
    m = re.match(r"\s+([A-Z_0-9]+)\s+([0-9.e+]+)\s*$", ln)
    if m:
        vals[m.group(1)] = float(m.group(2))
w, f = vals["WRITE_SIZE"], vals["FETCH_SIZE"]
j = {
    "note": "HBM traffic and VALU instructions of ONE launch of the tube kernel on this workload, from separate rocprofv3 --pmc passes "
            "(tools/profile_round.sh): traffic = WRITE_SIZE [KB] * 1024 + 2 * FETCH_SIZE [KB] * 1024 (gfx950 FETCH_SIZE counts half of wide "
            "coalesced reads, MI355X_MICROARCH.md HBM section)",
    "kernel_source_sha16": bench.kernel_source_hash(),
    "workload": {"voices_per_gpu": voices, "frames_per_voice": frames, "kind": kind, "kernel_form": form},
    "WRITE_SIZE_KB": w, "FETCH_SIZE_KB": f,
    "traffic_bytes_per_launch": int(w * 1024 + 2 * f * 1024),
    "SQ_INSTS_VALU": vals.get("SQ_INSTS_VALU"),
    "issue_cycles_per_valu": cyc,
    "issue_cycles_source": "mix-weighted issue cost of the kernel's instruction classes, profiles/valu_ceiling_r02.txt (tools/ubench/valu_ceiling.hip x tools/isa_mix.py)",
    "source": os.path.relpath(summary, ROOT) + " (separate --pmc passes)",
}
json.dump(j, open(out, "w"), indent=1)
print(json.dumps(j, indent=1))
