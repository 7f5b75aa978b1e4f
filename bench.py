#!/usr/bin/env python3
"""bench.py -- throughput of the Tube Resonance Model hot path on MI355X.

A "step" = one pass of -[TRMTubeModel synthesize] over one resident batch of synthetic control tracks.

  --gpus 1 (default)  BASELINE.json configs[1]: 4096 static-vowel tubes x 1 s @ 44.1 kHz, fp32.
  --gpus N > 1        BASELINE.json configs[4]: 8192 config-3 voices (time-varying gnuspeech.input tracks) per GPU,
                      i.e. 65 536 utterances over 8 GPUs; every rank runs its own shard, no data-path collective
                      (voices are independent): weak scaling.  One rank per GPU over torch.distributed (RCCL) for the
                      barrier and the MAX-over-ranks time.  When the ranks are not there yet (no WORLD_SIZE in the
                      environment), this process starts them as a CHILD `python -m torch.distributed.run` before it
                      touches the GPU and relays rank 0's line.

Prints ONE JSON line on rank 0 (DESIGN.md "Measurement" explains every field).
"""
import argparse
import hashlib
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
KERNEL_SOURCES = ("trm_oct.hip", "trm_oct.h", "trm_quad.hip", "trm_quad.h", "trm_quad_dev.h", "trm_kernels.hip", "trm_lane.h", "trm_devutil.h", "trm_kernels.h",
                  "Makefile")


def kernel_source_hash():
    """What a PMC profile under profiles/ is stamped with: the sources (and build flags) of the tube kernels."""
    h = hashlib.sha256()
    for f in KERNEL_SOURCES:
        h.update(open(os.path.join(ROOT, "gnuspeech_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


def usable_cores():
    """Host cores this process may actually use: the affinity mask, cut to the cgroup's CPU quota where there is one
    (a GPU box hands each job a share of its host, e.g. 16 of 256 hardware threads)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // per))
        except (OSError, ValueError):
            pass
    return n


def cpu_baseline(pd, frames, wall_s=3.0):
    """The oracle (CPU restatement of Frameworks/Tube, double precision: oracle/trm_oracle.c) timed on this host: one
    voice per task, in-process threads on all host cores (ctypes drops the GIL; the C library keeps no global state;
    one call per thread, so no Python runs inside the timed region).  The sample is the workload's own voices, gone
    through cyclically until every core has had about `wall_s` seconds of work."""
    import ctypes as C
    import threading
    import numpy as np
    import oracle_lib as O
    L = O.lib()
    cores = usable_cores()
    try:        # keep the per-voice buffers (350 KB) on the heap: mmap / munmap per voice serialises the threads in the kernel
        libc = C.CDLL(None)
        libc.mallopt(-3, 1 << 30)       # M_MMAP_THRESHOLD
        libc.mallopt(-1, 1 << 30)       # M_TRIM_THRESHOLD
    except (OSError, AttributeError):
        pass
    fr = np.ascontiguousarray(np.asarray(frames, dtype=np.float32).astype(np.float64))     # what the GPU path is handed
    nv, nf = fr.shape[0], fr.shape[1]
    op = O.InputParams.from_dict(pd)
    fp = fr.ctypes.data_as(C.POINTER(C.c_double))

    def run(first, count):
        n = C.c_uint64()
        rc = L.trm_oracle_run_voices(C.byref(op), fp, nf, nv, first, count, C.byref(n))
        assert rc == 0, rc
        return n.value
    t0 = time.perf_counter()
    per_voice_samples = run(0, 1)
    per_voice = max(time.perf_counter() - t0, 1e-4)
    per_thread = max(1, int(wall_s / per_voice))
    res = [0] * cores
    go = threading.Barrier(cores + 1)

    def worker(i):
        go.wait()                                   # every thread exists before the clock starts
        res[i] = run((i * per_thread) % nv, per_thread)
        go.wait()
    th = [threading.Thread(target=worker, args=(i,)) for i in range(cores)]
    for t in th:
        t.start()
    go.wait()
    t0 = time.perf_counter()
    go.wait()
    dt = time.perf_counter() - t0
    for t in th:
        t.join()
    return {"value": float(sum(res)) / dt, "unit": "samples/s", "cores": cores, "kind": "port",
            "per_core": float(sum(res)) / dt / cores,
            "host_hardware_threads": os.cpu_count(),
            "sample": "%d voice runs (the workload's %d voices cyclically, %d output samples each) = %.0f s of CPU work in "
                      "%.2f s on %d threads (the cores this job may use), oracle/trm_oracle.c (double), one voice per task"
                      % (cores * per_thread, nv, per_voice_samples, dt * cores, dt, cores)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)        # (a step is 2.5 ms: the default run is the CPU baseline's 3 s + 65 ms)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--voices", type=int, default=None, help="voices per GPU (default: 4096 at --gpus 1 = configs[1]; 8192 at --gpus N = configs[4])")
    ap.add_argument("--seconds", type=float, default=1.0)
    ap.add_argument("--workload", default=None, choices=["static", "timevarying"],
                    help="default: static at --gpus 1 (configs[1]), timevarying at --gpus N (configs[4]: config-3 voices)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--kernel", default="auto", choices=["auto", "wide", "quad", "oct"],
                    help="kernel form (include/trm_c_api.h); auto = the library's choice by batch size")
    a = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        # No ranks yet: start them as a child process (never re-exec a process that may have touched the GPU -- this one
        # has not even imported torch) and pass rank 0's line through.
        import socket
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        sys.exit("bench.py: --gpus %d but WORLD_SIZE=%d: launch with --nproc-per-node equal to --gpus" % (a.gpus, world))
    dist = world > 1
    voices = a.voices if a.voices is not None else (8192 if dist else 4096)
    workload = a.workload if a.workload is not None else ("timevarying" if dist else "static")

    import numpy as np
    import cases
    pd = cases.monet_default_params(44100.0)
    nframes = int(round(a.seconds * 250)) + 1
    # per-rank shard: independent voices, different seed offset per rank (no data-path collective)
    if workload == "static":
        frames = cases.config2_frames(voices, nframes=nframes, seed=20250117 + rank)
        wname = "configs[1]: batch=%d static-vowel tubes x %.3g s @ 44.1 kHz, Monet default voice, fp32" % (voices, a.seconds)
    else:
        frames = cases.config3_frames(voices, nframes=nframes, seed=20250118 + rank)
        wname = "configs[2]: batch=%d time-varying tubes (gnuspeech.input tracks) x %.3g s @ 44.1 kHz" % (voices, a.seconds)
    if dist:
        wname = "configs[4]: %d utterances sharded %d per GPU over %d GPUs (no collective); per GPU = %s" % (voices * world, voices, world, wname)

    # the CPU leg first: before this process has a GPU context (rank 0 at N=1 only)
    cpu = None
    if rank == 0 and not dist and not a.no_cpu_baseline:
        cpu = cpu_baseline(pd, frames)

    import torch
    import gnuspeech_amd as g
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
    # roofline of the dominant kernel (trm_tube_kernel*): algorithmic bytes per launch =
    # 4 B x output samples + 64 B x frames, / its average launch duration
    alg_bytes = 4.0 * samples_per_step_rank + 64.0 * voices * nframes
    avg_launch_s = (kern_ms / max(1, launches)) * 1e-3
    achieved = alg_bytes / avg_launch_s / 1e9
    # HBM bytes per launch come from separate rocprofv3 --pmc passes (they cannot be combined with the timed run);
    # the committed file is only quoted when it was collected on THIS workload with THESE kernel sources
    # (tools/make_traffic.py stamps it with kernel_source_hash()), otherwise traffic is null
    traffic = None
    valu = None
    traffic_note = "no PMC profile for this workload"
    try:
        tj = json.load(open(os.path.join(ROOT, "profiles", "traffic_r02.json")))
        w = tj["workload"]
        if tj.get("kernel_source_sha16") != kernel_source_hash():
            traffic_note = "profiles/traffic_r02.json is stale (kernel sources changed since the PMC pass): not quoted"
        elif (w["voices_per_gpu"], w["frames_per_voice"], w["kind"], w["kernel_form"]) == (voices, nframes, workload, b.last_kernel):
            traffic = tj["traffic_bytes_per_launch"]
            traffic_note = tj["source"]
            if "SQ_INSTS_VALU" in tj and "issue_cycles_per_valu" in tj:
                # wave64 VALU instructions of one launch (PMC) x the mix-weighted issue cost measured by tools/ubench
                # (profiles/valu_ceiling_r02.txt) against this run's launch time on 1024 SIMDs at the 2.4 GHz peak clock
                valu = {"insts_per_launch": tj["SQ_INSTS_VALU"], "issue_cycles_per_inst": tj["issue_cycles_per_valu"],
                        "source": tj["source"] + "; " + tj.get("issue_cycles_source", ""),
                        "frac_of_issue_slots": tj["SQ_INSTS_VALU"] * tj["issue_cycles_per_valu"] / (avg_launch_s * 2.4e9 * 1024)}
    except (OSError, KeyError, ValueError):
        pass
    out = {
        "metric": "audio samples/s (whole node) + concurrent real-time tube voices",
        "value": value, "unit": "samples/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "realtime_voices_44k1": value / 44100.0,
        "config": {"workload": wname, "voices_per_gpu": voices, "frames_per_voice": nframes,
                   "output_samples_per_voice": samples_per_step_rank // max(1, voices),
                   "tube_rate_hz": b.derived["sampleRate"], "control_rate_hz": 250, "sharding": "voices, no collective",
                   "kernel_form": b.last_kernel},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_note,
                     "kernel": {"wide": "trm_tube_kernel", "quad": "trm_tube_kernel_q", "oct": "trm_tube_kernel_o"}[b.last_kernel], "avg_launch_ms": kern_ms / max(1, launches),
                     "algorithmic_bytes_per_launch": alg_bytes, "valu_issue": valu,
                     "note": "VALU-issue bound scalar recurrence (SURVEY 8d); HBM-write fraction reported as BASELINE asks"},
        "cpu_baseline": cpu,
    }
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist:
        td.destroy_process_group()


if __name__ == "__main__":
    main()
