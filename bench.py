#!/usr/bin/env python3
"""bench.py -- throughput of the Tube Resonance Model hot path on MI355X.

A "step" = one pass of -[TRMTubeModel synthesize] over one resident batch of synthetic control
tracks.  N=1 workload = BASELINE.json configs[1]: 4096 static-vowel tubes x 1 s @ 44.1 kHz, fp32,
four lanes per tube (the library picks the kernel form by batch size).  N>1: the same per-GPU batch on every rank (weak scaling, no collective: voices
are independent), launched one rank per GPU by torch.distributed.run.

Prints ONE JSON line on rank 0 (see DESIGN.md "Measurement" for every field).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def cpu_baseline(pd, frames, budget_s=12.0):
    """The oracle (CPU restatement of Frameworks/Tube, double precision) timed on this host, one
    voice per process over all cores, on a bounded sample of the same workload."""
    import multiprocessing as mp
    import numpy as np
    cores = os.cpu_count() or 1
    t0 = time.time()
    r = _cpu_one((pd, np.asarray(frames[0], dtype=np.float64)))        # one voice: how long is it?
    per_voice = max(time.time() - t0, 1e-3)
    nv = int(max(cores, min(len(frames), cores * max(1, int(budget_s / per_voice)))))
    nv = min(nv, len(frames))
    work = [(pd, np.asarray(frames[v], dtype=np.float64)) for v in range(nv)]
    t0 = time.time()
    with mp.get_context("fork").Pool(cores) as pool:
        ns = pool.map(_cpu_one, work, chunksize=max(1, nv // (cores * 4)))
    dt = time.time() - t0
    return {"value": float(sum(ns)) / dt, "unit": "samples/s", "cores": cores, "kind": "port",
            "sample": "%d of the workload's voices x %d output samples each, oracle/trm_oracle.c (double), "
                      "one voice per process on %d cores, %.1f s" % (nv, r, cores, dt)}


def _cpu_one(args):
    import numpy as np
    import oracle_lib as O
    pd, fr = args
    o = O.synthesize(O.InputParams.from_dict(pd), np.asarray(fr, dtype=np.float32).astype(np.float64))
    return o["numberSamples"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--voices", type=int, default=4096, help="voices per GPU (BASELINE configs[1]: 4096)")
    ap.add_argument("--seconds", type=float, default=1.0)
    ap.add_argument("--workload", default="static", choices=["static", "timevarying"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--kernel", default="auto", choices=["auto", "wide", "quad"],
                    help="kernel form (include/trm_c_api.h); auto = the library's choice by batch size")
    a = ap.parse_args()

    import numpy as np
    import torch
    import cases
    import gnuspeech_amd as g

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = world > 1
    # rehearsal on a box with fewer GPUs than ranks (never the measured configuration): TRM_BENCH_REHEARSAL=1 puts every
    # rank on GPU 0 and uses gloo for the barrier / MAX reduction -- same control flow, no RCCL
    rehearsal = os.environ.get("TRM_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if dist:
        import torch.distributed as td
        if rehearsal:
            td.init_process_group(backend="gloo")
        else:
            td.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))

    pd = cases.monet_default_params(44100.0)
    nframes = int(round(a.seconds * 250)) + 1
    # per-rank shard: independent voices, different seed offset per rank (no data-path collective)
    if a.workload == "static":
        frames = cases.config2_frames(a.voices, nframes=nframes, seed=20250117 + rank)
        wname = "configs[1]: batch=%d static-vowel tubes x %.3g s @ 44.1 kHz, Monet default voice, fp32" % (a.voices, a.seconds)
    else:
        frames = cases.config3_frames(a.voices, nframes=nframes, seed=20250118 + rank)
        wname = "configs[2]: batch=%d time-varying tubes (gnuspeech.input tracks) x %.3g s @ 44.1 kHz" % (a.voices, a.seconds)

    b = g.TRMBatch(g.TRMInputParameters.from_dict(pd), device=local_rank)
    b.set_kernel(a.kernel)
    st = b.prepare_device(frames, device="cuda:%d" % local_rank)
    stream = torch.cuda.current_stream()

    for _ in range(a.warmup):
        b.synthesize_device(st, stream)
    torch.cuda.synchronize()
    b.kernel_time_ms()                                     # reset the per-launch event accumulator
    if dist:
        td.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        b.synthesize_device(st, stream)
    torch.cuda.synchronize()
    if dist:
        td.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if dist:
        tmax = torch.tensor([dt], dtype=torch.float64, device="cpu" if rehearsal else "cuda")
        td.all_reduce(tmax, op=td.ReduceOp.MAX)
        dt = float(tmax.item())
    kern_ms, launches = b.kernel_time_ms()                 # hipEvents on the launch stream

    samples_per_step_rank = int(st["total_out"])
    total_samples = samples_per_step_rank * a.steps * world
    value = total_samples / dt
    # roofline of the dominant kernel (trm_tube_kernel): algorithmic bytes per launch =
    # 4 B x output samples + 64 B x frames, / its average launch duration
    alg_bytes = 4.0 * samples_per_step_rank + 64.0 * a.voices * nframes
    avg_launch_s = (kern_ms / max(1, launches)) * 1e-3
    achieved = alg_bytes / avg_launch_s / 1e9
    # HBM bytes per launch from the PMC passes kept under profiles/ (collected separately: --pmc cannot be
    # combined with the timed run); only quoted when it was measured on this very workload
    traffic = None
    valu = None
    try:
        tj = json.load(open(os.path.join(ROOT, "profiles", "traffic_r01.json")))
        w = tj["workload"]
        if (w["voices_per_gpu"], w["frames_per_voice"], w["kind"]) == (a.voices, nframes, a.workload):
            traffic = tj["traffic_bytes_per_launch"]
            if b.last_kernel == "quad" and "SQ_INSTS_VALU" in tj:
                # the bound that binds: wave64 VALU instructions (PMC, same profile) x 4 cycles on one
                # of 256 CUs x 4 SIMD16, against this run's launch time at the 2.4 GHz peak clock
                valu = {"insts_per_launch": tj["SQ_INSTS_VALU"], "source": "profiles/traffic_r01.json (rocprofv3 --pmc SQ_INSTS_VALU)",
                        "frac_of_issue_slots": tj["SQ_INSTS_VALU"] * 4.0 / (avg_launch_s * 2.4e9 * 1024)}
    except (OSError, KeyError, ValueError):
        pass
    out = {
        "metric": "audio samples/s (whole node) + concurrent real-time tube voices",
        "value": value, "unit": "samples/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "realtime_voices_44k1": value / 44100.0,
        "config": {"workload": wname, "voices_per_gpu": a.voices, "frames_per_voice": nframes,
                   "output_samples_per_voice": samples_per_step_rank // max(1, a.voices),
                   "tube_rate_hz": b.derived["sampleRate"], "control_rate_hz": 250, "sharding": "voices, no collective",
                   "kernel_form": b.last_kernel},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                     "kernel": {"wide": "trm_tube_kernel", "quad": "trm_tube_kernel_q"}[b.last_kernel], "avg_launch_ms": kern_ms / max(1, launches),
                     "algorithmic_bytes_per_launch": alg_bytes, "valu_issue": valu,
                     "note": "VALU-issue bound scalar recurrence (SURVEY 8d); HBM-write fraction reported as BASELINE asks"},
    }
    if rank == 0:
        if not a.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(pd, frames)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out))
    if dist:
        td.destroy_process_group()


if __name__ == "__main__":
    main()
