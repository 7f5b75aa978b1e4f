#!/usr/bin/env python3
"""Static instruction counts of ONE role of trm_oct.hip's kernel: the kernel compiled with every wave forced into that role
(-DTRM_EXPERIMENTS -DTRM_ISA_ROLE=n: the other role bodies fold away), the role's step loop found as the loop that holds the
step barrier, its basic blocks listed with their instruction counts by kind.  Blocks that run once per control period or per
converter block (period set-up, row staging, flush) are part of the listing: read the per-block numbers, not only the sum.
usage: isa_role.py <role 0..5: osc mix area fric tube convert> [-v]"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
role = int(sys.argv[1]); verbose = "-v" in sys.argv
with tempfile.TemporaryDirectory() as d:
    out = os.path.join(d, "k.s")
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "--offload-device-only", "-S", "-o", out, "-O3", "-std=c++17", "-fno-slp-vectorize",
                           "-mllvm", "-amdgpu-sched-strategy=iterative-ilp", "-DTRM_EXPERIMENTS", "-DTRM_ISA_ROLE=%d" % role,
                           os.path.join(ROOT, "gnuspeech_amd", "csrc", "trm_oct.hip")], stderr=subprocess.DEVNULL)
    lines = open(out).read().split("\n")
# basic blocks: label line or "; %bb.N:" comment starts one; its loop membership is in the comment
blocks = []; cur = None
for i, l in enumerate(lines):
    m = re.match(r"^(\.LBB\d+_\d+):\s*(;.*)?$", l) or re.match(r"^; %bb\.(\d+):\s*(;.*)?$", l)
    if m:
        cur = {"name": m.group(1), "note": (m.group(2) or ""), "ins": []}
        blocks.append(cur)
        continue
    t = l.strip()
    if cur is not None and t and not t.startswith((";", ".", "//")) and not t.endswith(":"):
        cur["ins"].append(t)
# the step loop: the loop header named by the block that holds the LAST s_barrier of the file
hold = [b for b in blocks if "s_barrier" in b["ins"]]
last = hold[-1]
m = re.search(r"Header=(BB\d+_\d+)", last["note"])
hdr = m.group(1) if m else last["name"].lstrip(".L")
inloop = [b for b in blocks if ("Header=" + hdr) in b["note"] or b["name"] == ".L" + hdr]
def kinds(ins):
    v = sum(1 for t in ins if t.startswith("v_")); s = sum(1 for t in ins if t.startswith("s_") and not t.startswith(("s_waitcnt", "s_nop")))
    d = sum(1 for t in ins if t.startswith("ds_")); g = sum(1 for t in ins if t.startswith(("global_", "buffer_", "flat_")))
    return v, s, d, g
tot = [0, 0, 0, 0]
print("role %d: step loop %s, %d blocks" % (role, hdr, len(inloop)))
for b in inloop:
    k = kinds(b["ins"])
    for i in range(4): tot[i] += k[i]
    if verbose or k[0] >= 8:
        print("  %-12s VALU %4d SALU %4d LDS %3d VMEM %2d   %s" % (b["name"], k[0], k[1], k[2], k[3], b["note"].strip()[:60]))
print("  all blocks: VALU %d SALU %d LDS %d VMEM %d" % tuple(tot))
