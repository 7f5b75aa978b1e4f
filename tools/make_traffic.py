#!/usr/bin/env python3
"""Adds the PMC passes of one profiled workload (tools/profile_workload.sh -> summary.txt) to profiles/traffic_r04.json --
the file bench.py quotes `roofline.traffic` and `roofline.valu_issue` from.  The file is a LIST of entries keyed by
(voices per GPU, frames per voice, kind, kernel form) and is STAMPED as a whole with the hash of the kernel sources
(bench.kernel_source_hash()): bench.py refuses a file whose stamp differs from the sources it is running, and this tool
drops the entries of another stamp when it adds one.
HBM bytes per launch as MI355X_MICROARCH.md prescribes: WRITE_SIZE and FETCH_SIZE collected in SEPARATE passes, both in
KB, FETCH_SIZE doubled on gfx950 (it counts half of wide coalesced reads).
`issue cycles per VALU` = `mix`: bench.ISSUE_CYCLES of the kernel form -- the static instruction mix of the kernel (tools/isa_mix.py) priced
with the per-class issue costs of tools/ubench/valu_ceiling.hip.  (Round 3 priced with 4 x SQ_ACTIVE_INST_VALU / SQ_INSTS_VALU of the
pass itself; the calibration of round 4, profiles/valu_pmc_calibration_r04.txt, showed that counter to charge one quad-cycle per
instruction whatever it costs the SIMD: it is an instruction count.)
usage: make_traffic.py <summary.txt> <voices> <frames> <static|timevarying|ragged> <oct|quad|wide|wide/split> <issue cycles per VALU | mix> [out.json]"""
import json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench

summary, voices, frames, kind, form, cyc = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], sys.argv[5], sys.argv[6]
out = sys.argv[7] if len(sys.argv) > 7 else bench.TRAFFIC_FILE
vals = {}
for ln in open(summary):
    m = re.match(r"\s+([A-Z_0-9]+)\s+([0-9.e+]+)\s*$", ln)
    if m:
        vals[m.group(1)] = float(m.group(2))
w, f = vals["WRITE_SIZE"], vals["FETCH_SIZE"]
cyc = bench.ISSUE_CYCLES[form] if cyc in ("mix", "auto") else float(cyc)
entry = {
    "workload": {"voices_per_gpu": voices, "frames_per_voice": frames, "kind": kind, "kernel_form": form},
    "WRITE_SIZE_KB": w, "FETCH_SIZE_KB": f,
    "traffic_bytes_per_launch": int(w * 1024 + 2 * f * 1024),
    "SQ_INSTS_VALU": vals.get("SQ_INSTS_VALU"), "SQ_INSTS_SALU": vals.get("SQ_INSTS_SALU"),
    "SQ_WAIT_ANY_over_WAVE_CYCLES": (vals["SQ_WAIT_ANY"] / vals["SQ_WAVE_CYCLES"]) if vals.get("SQ_WAVE_CYCLES") else None,
    "SQ_LDS_BANK_CONFLICT_over_IDX_ACTIVE": (vals["SQ_LDS_BANK_CONFLICT"] / vals["SQ_LDS_IDX_ACTIVE"]) if vals.get("SQ_LDS_IDX_ACTIVE") else None,
    "issue_cycles_per_valu": cyc,
    "SQ_ACTIVE_INST_VALU": vals.get("SQ_ACTIVE_INST_VALU"),
    "SQ_WAIT_INST_ANY_over_WAVE_CYCLES": (vals["SQ_WAIT_INST_ANY"] / vals["SQ_WAVE_CYCLES"]) if vals.get("SQ_WAVE_CYCLES") and "SQ_WAIT_INST_ANY" in vals else None,
    "issue_cycles_source": "the kernel's static instruction mix (tools/isa_mix.py) priced per class: plain VGPR-operand VALU 2.4 cycles on a shared SIMD, packed / fp64 / DPP / compare-select / scalar-operand 4.3, transcendental 8.2 (tools/ubench/valu_ceiling.hip; profiles/valu_pmc_calibration_r04.txt)",
    "source": os.path.relpath(summary, ROOT) + " (separate --pmc passes)",
}
stamp = bench.kernel_source_hash()
try:
    tj = json.load(open(out))
    if tj.get("kernel_source_sha16") != stamp:
        tj = None
except (OSError, ValueError):
    tj = None
if tj is None:
    tj = {"note": "HBM traffic and VALU instructions of ONE launch of the tube kernel per profiled workload, from separate rocprofv3 --pmc "
                  "passes (tools/profile_workload.sh): traffic = WRITE_SIZE [KB] * 1024 + 2 * FETCH_SIZE [KB] * 1024 (gfx950 FETCH_SIZE counts "
                  "half of wide coalesced reads, MI355X_MICROARCH.md HBM section)",
          "kernel_source_sha16": stamp, "entries": []}
tj["entries"] = [e for e in tj["entries"] if e["workload"] != entry["workload"]] + [entry]
json.dump(tj, open(out, "w"), indent=1)
print(json.dumps(entry, indent=1))
