#!/usr/bin/env python3
"""Condenses rocprofv3 CSV output (kernel stats + optional PMC passes) into one small text file for profiles/.

A batch of more than 65 536 voices of the one-voice-per-lane kernel goes out as several dispatches per launch (slices of 1024
workgroups, trm_kernels.hip launch_tube): TRM_SUMMARY_DISPATCHES=N sums the counters of the last N tube-kernel dispatches
(= one launch) instead of the last one.
usage: rocprof_summary.py <out.txt> <kernel_trace_dir> [<pmc_dir> ...]"""
import collections
import csv
import glob
import os
import sys


def main():
    out, trace_dir, pmc_dirs = sys.argv[1], sys.argv[2], sys.argv[3:]
    lines = []
    for f in glob.glob(trace_dir + "/**/*_kernel_stats.csv", recursive=True):
        lines.append("== rocprofv3 --kernel-trace --stats (%s)" % f)
        lines += [ln.rstrip() for ln in open(f)]
    for f in glob.glob(trace_dir + "/**/*_kernel_trace.csv", recursive=True):
        rows = [r for r in csv.DictReader(open(f)) if "trm_tube_kernel" in r["Kernel_Name"]]
        if rows:
            r = rows[-1]
            # rocprofv3 writes the geometry per dimension (Grid_Size_X = threads, not workgroups)
            gx, wx = int(r.get("Grid_Size_X", 0)), int(r.get("Workgroup_Size_X", 0) or 1)
            lines.append("== last trm_tube_kernel dispatch (%s): grid %d threads = %d workgroups of %d, VGPR %s SGPR %s LDS %s scratch %s" % (
                r["Kernel_Name"].split("(")[0][-40:], gx, gx // wx, wx, r.get("VGPR_Count"), r.get("SGPR_Count"),
                r.get("LDS_Block_Size"), r.get("Scratch_Size")))
            d = [int(x["End_Timestamp"]) - int(x["Start_Timestamp"]) for x in rows]
            lines.append("   dispatch durations ns: n=%d min=%d median=%d max=%d" % (len(d), min(d), sorted(d)[len(d) // 2], max(d)))
    for pd in pmc_dirs:
        for f in glob.glob(pd + "/**/*_counter_collection.csv", recursive=True):
            agg = collections.defaultdict(float)
            for r in csv.DictReader(open(f)):
                if "trm_tube_kernel" in r["Kernel_Name"]:
                    agg[(int(r["Dispatch_Id"]), r["Counter_Name"])] += float(r["Counter_Value"])
            if not agg:
                continue
            nd = int(os.environ.get("TRM_SUMMARY_DISPATCHES", "1"))
            ids = sorted(set(k[0] for k in agg))[-nd:]
            lines.append("== rocprofv3 --pmc (%s), last %d trm_tube_kernel dispatch(es) (ids %s) = one launch, summed over XCDs/SEs" % (f, nd, ids))
            tot = collections.defaultdict(float)
            for (d, name), v in agg.items():
                if d in ids:
                    tot[name] += v
            for name, v in sorted(tot.items()):
                lines.append("   %-28s %.6g" % (name, v))
    open(out, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()
