#!/usr/bin/env python3
"""Joins tools/ubench/valu_ceiling's own table (cycles per instruction per SIMD for W = 1..4 waves per SIMD) with the PMC
rows rocprofv3 collected for the same run.  Every (class, W) is launched twice (warm-up, measurement): the second dispatch of
each pair is used.  Prints, per class and W:
    cyc/inst/SIMD     the benchmark's own in-kernel figure (shader cycles per wave64 instruction per SIMD)
    4*ACTIVE/SIMDcyc  4 x SQ_ACTIVE_INST_VALU / (4 SIMDs x kernel cycles): above 1 the counter cannot be "cycles the SIMD's VALU is busy"
    ACTIVE/INSTS      quad-cycles the counter charges per instruction
usage: valu_pmc_join.py <under_pmc.txt> <pmc_dir>"""
import collections, csv, glob, re, sys

table, pmc_dir = sys.argv[1], sys.argv[2]
rows = []
for ln in open(table):
    m = re.match(r"(.{44})((?:\s+W=\d\s+[0-9.]+)+)\s*$", ln)
    if m:
        rows.append((m.group(1).strip(), [float(x) for x in re.findall(r"W=\d\s+([0-9.]+)", m.group(2))]))
agg = collections.defaultdict(dict)
order = []
for f in glob.glob(pmc_dir + "/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "void k<" not in r["Kernel_Name"]:       # (the runtime's own copy kernels between the launches)
            continue
        d = int(r["Dispatch_Id"])
        if d not in agg:
            order.append(d)
        agg[d][r["Counter_Name"]] = agg[d].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
order.sort()
N = 256 * 16 * 8          # ITERS x REP x 8 instructions per wave (valu_ceiling.hip)
print("%-44s %2s %13s %16s %12s %14s %12s" % ("class", "W", "cyc/inst/SIMD", "4*ACTIVE/SIMDcyc", "ACTIVE/INSTS", "WAIT_INST/WAVE", "INSTS_VALU"))
i = 0
for name, vals in rows:
    for w, cyc in enumerate(vals, start=1):
        if i + 1 >= len(order):
            break
        c = agg[order[i + 1]]          # the measured launch of the pair
        i += 2
        kernel_cycles = cyc * N * w    # per SIMD: N instructions from each of its w waves
        act, insts = c.get("SQ_ACTIVE_INST_VALU", 0.0), c.get("SQ_INSTS_VALU", 0.0)
        print("%-44s %2d %13.2f %16.3f %12.3f %14.3f %12.0f" % (name, w, cyc, 4.0 * act / (4.0 * kernel_cycles) if kernel_cycles else 0, act / insts if insts else 0,
                                                        c.get("SQ_WAIT_INST_ANY", 0.0) / c["SQ_WAVE_CYCLES"] if c.get("SQ_WAVE_CYCLES") else 0, insts))
