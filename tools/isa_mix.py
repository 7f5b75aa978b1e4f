#!/usr/bin/env python3
"""Static instruction mix of the tube kernels by ISSUE-COST CLASS (tools/ubench/valu_ceiling.hip measured the classes):
  fast   plain VALU, VGPR operands only                         2.4 cycles per SIMD with >= 2 waves, 4.8 for a lone wave
  slow   packed (v_pk_*), fp64, DPP, any SGPR / VCC operand,    4.3 cycles whatever the number of waves (4.9 lone)
         v_cmp*, v_cndmask*
  trans  v_rcp / v_exp / v_log / v_sqrt / v_rsq / v_sin / v_cos 8.2
Compiles the kernel source to assembly with the product's flags and splits each kernel at its step barriers: the piece
in front of a barrier that lies inside a loop is one role's per-step body (period-boundary blocks included: they run
once per control period, so the per-step figures are slight over-counts).
usage: isa_mix.py [quad|oct|wide]"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
which = sys.argv[1] if len(sys.argv) > 1 else "quad"
src = {"quad": "trm_quad.hip", "oct": "trm_oct.hip", "wide": "trm_kernels.hip"}[which]
flags = ["-O3", "-std=c++17", "-fno-slp-vectorize"] + (["-mllvm", "-amdgpu-sched-strategy=iterative-ilp"] if which != "wide" else [])
with tempfile.TemporaryDirectory() as d:
    out = os.path.join(d, "k.s")
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "--offload-device-only", "-S", "-o", out] + flags + [os.path.join(ROOT, "gnuspeech_amd", "csrc", src)],
                          stderr=subprocess.DEVNULL)
    text = open(out).read()
TRANS = re.compile(r"^v_(rcp|exp|log|sqrt|rsq|sin|cos)_")
SGPR = re.compile(r"(?<![a-z_0-9])(s\d+|s\[\d+:\d+\]|vcc|exec)\b")


def classify(line):
    op = line.split()[0]
    if not op.startswith("v_"):
        return "salu" if op.startswith("s_") else ("lds" if op.startswith("ds_") else ("vmem" if op.startswith(("global_", "buffer_", "flat_")) else "other"))
    if TRANS.match(op):
        return "trans"
    rest = line[len(op):]
    if op.startswith("v_pk_") or "_f64" in op or "_dpp" in op or " row_" in rest or "quad_perm" in rest or op.startswith(("v_cmp", "v_cndmask", "v_readlane", "v_readfirstlane", "v_writelane")):
        return "slow"
    # the destination is the first operand; SGPRs among the SOURCES make it a slow-class issue
    ops = rest.split(",")
    if any(SGPR.search(o) for o in ops[1:]):
        return "slow"
    return "fast"


for m in re.finditer(r"^(_ZN3trm\w+):.*?\n(.*?)\n\s+s_endpgm", text, re.S | re.M):
    name, body = m.group(1), m.group(2).split("\n")
    if "tube_kernel" not in name:
        continue
    print("== %s" % name)
    # role bodies: from the loop header label whose back edge follows the barrier, to the barrier
    barr = [i for i, l in enumerate(body) if l.strip() == "s_barrier"]
    total = {}
    for bi in barr:
        # the enclosing loop header: the nearest "; =>This Inner Loop Header" label above the barrier
        hdr = next((i for i in range(bi, -1, -1) if "Loop Header" in body[i] and "Depth=1" in body[i]), None)
        if hdr is None:
            continue
        prev_barr = max([b for b in barr if b < bi] + [-1])
        if hdr < prev_barr:
            continue
        cnt = {}
        sub = {}
        for l in body[hdr:bi]:
            t = l.strip()
            if not t or t.startswith((";", ".", "//")) or t.endswith(":"):
                continue
            c = classify(t)
            cnt[c] = cnt.get(c, 0) + 1
            if c == "slow":
                op = t.split()[0]
                k = "packed" if op.startswith("v_pk_") else "fp64" if "_f64" in op else "dpp" if ("_dpp" in op or "row_" in t or "quad_perm" in t) else "cmp/cndmask" if op.startswith(("v_cmp", "v_cndmask")) else "sgpr operand"
                sub[k] = sub.get(k, 0) + 1
        valu = cnt.get("fast", 0) + cnt.get("slow", 0) + cnt.get("trans", 0)
        if valu < 20:
            continue
        print("  step body at line %5d: VALU %4d = fast %4d + slow %4d (%s) + trans %3d | LDS %3d SALU %3d VMEM %2d | issue cycles lone wave %5.0f, shared SIMD %5.0f" % (
            hdr, valu, cnt.get("fast", 0), cnt.get("slow", 0), ", ".join("%s %d" % kv for kv in sorted(sub.items())), cnt.get("trans", 0),
            cnt.get("lds", 0), cnt.get("salu", 0), cnt.get("vmem", 0),
            4.85 * (cnt.get("fast", 0) + cnt.get("slow", 0)) + 8.6 * cnt.get("trans", 0),
            2.4 * cnt.get("fast", 0) + 4.3 * cnt.get("slow", 0) + 8.2 * cnt.get("trans", 0)))
        for k, v in cnt.items():
            total[k] = total.get(k, 0) + v
    v = total.get("fast", 0) + total.get("slow", 0) + total.get("trans", 0)
    if v:
        print("  all roles, one step: VALU %d: fast %.0f %%, slow %.0f %%, trans %.1f %%  -> mix-weighted issue cost %.2f cycles per VALU instruction on a shared SIMD, %.2f for lone waves"
              % (v, 100.0 * total.get("fast", 0) / v, 100.0 * total.get("slow", 0) / v, 100.0 * total.get("trans", 0) / v,
                 (2.4 * total.get("fast", 0) + 4.3 * total.get("slow", 0) + 8.2 * total.get("trans", 0)) / v,
                 (4.85 * (total.get("fast", 0) + total.get("slow", 0)) + 8.6 * total.get("trans", 0)) / v))
