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
  --config K          the per-GPU workload by BASELINE.json configs[] index, whatever --gpus says: 1 = 4096 static
                      tubes, 2 = 4096 time-varying tubes, 3 = the GnuTTSServer sentence batch (1024 ragged utterances of
                      0.6 - 6 s, seed 20250119; `value` = the batch resident on the device, `end_to_end` = the same batch
                      through the C-ABI host entry: H2D + kernels + D2H, fp32 and int16), 4 = ONE GPU's shard of
                      configs[4] (8192 time-varying voices).  `--gpus 1 --config 4` is the single-GPU run an N-rank line
                      compares with (its `scaling_baseline`).

The default line (N = 1, configs[1]) also carries a `stream` record: 1 048 576 voices streamed in ten 100 ms chunks through
trm_stream_push_device (the metric's "concurrent real-time voices" half: every chunk must beat its audio time); --no-stream
leaves it out.

The N-rank line is self-contained for weak scaling: besides the contract's MAX-over-ranks time it carries every rank's
own elapsed and kernel time (`per_rank_ms`, `per_rank_kernel_ms`, gathered after the timed region), `per_gpu_value`, and
`solo_ms_per_step` = rank 0 running the same shard ALONE (the other ranks idle at a barrier; outside the timed region).

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


def cpu_baseline_ragged(pd, voices, wall_s=3.0):
    """cpu_baseline() for a ragged batch (configs[3]): one oracle call per voice (the voices differ in length), every thread
    works through its own share of the batch's voices, cyclically, until about `wall_s` seconds have passed."""
    import ctypes as C
    import threading
    import numpy as np
    import oracle_lib as O
    L = O.lib()
    cores = usable_cores()
    op = O.InputParams.from_dict(pd)
    arrs = [np.ascontiguousarray(np.asarray(v, dtype=np.float32).astype(np.float64)) for v in voices]

    def run(i):
        n = C.c_uint64()
        a = arrs[i % len(arrs)]
        rc = L.trm_oracle_run_voices(C.byref(op), a.ctypes.data_as(C.POINTER(C.c_double)), a.shape[0], 1, 0, 1, C.byref(n))
        assert rc == 0, rc
        return n.value
    res, cnt = [0] * cores, [0] * cores
    go = threading.Barrier(cores + 1)
    stop = [0.0]

    def worker(t):
        go.wait()
        i = t
        while time.perf_counter() < stop[0]:
            res[t] += run(i)
            cnt[t] += 1
            i += cores
        go.wait()
    th = [threading.Thread(target=worker, args=(t,)) for t in range(cores)]
    for t in th:
        t.start()
    stop[0] = time.perf_counter() + wall_s
    go.wait()
    t0 = time.perf_counter()
    go.wait()
    dt = time.perf_counter() - t0
    for t in th:
        t.join()
    return {"value": float(sum(res)) / dt, "unit": "samples/s", "cores": cores, "kind": "port",
            "per_core": float(sum(res)) / dt / cores, "host_hardware_threads": os.cpu_count(),
            "sample": "%d voice runs (the batch's %d utterances of %d - %d frames, every thread its own share, cyclically) = %.0f s of CPU "
                      "work in %.2f s on %d threads (the cores this job may use: %d of the host's %d hardware threads), "
                      "oracle/trm_oracle.c (double), one voice per task"
                      % (sum(cnt), len(arrs), min(len(a) for a in arrs), max(len(a) for a in arrs), dt * cores, dt, cores, cores,
                         os.cpu_count() or cores)}


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
                      "%.2f s on %d threads (the cores this job may use: %d of the host's %d hardware threads), "
                      "oracle/trm_oracle.c (double), one voice per task"
                      % (cores * per_thread, nv, per_voice_samples, dt * cores, dt, cores, cores, os.cpu_count() or cores)}


TRAFFIC_FILE = os.path.join(ROOT, "profiles", "traffic_r04.json")
# BASELINE.json configs[K] per GPU: (voices, kind)
CONFIGS = {1: (4096, "static"), 2: (4096, "timevarying"), 3: (1024, "ragged"), 4: (8192, "timevarying")}
# What a wave64 VALU instruction of each kernel costs a SIMD that several waves share, weighted by the kernel's static
# instruction mix (tools/isa_mix.py: plain VGPR-operand fp32 / integer instructions 2.4 cycles; packed, fp64, DPP,
# compare / select and scalar-operand ones 4.3; transcendentals 8.2 -- tools/ubench/valu_ceiling.hip).  Round 4 put the
# microbenchmark under the counters (profiles/valu_pmc_calibration_r04.txt): SQ_ACTIVE_INST_VALU charges ONE quad-cycle per
# instruction of any class (two per transcendental), i.e. it counts instructions, not busy SIMD cycles -- round 3's
# "4 x SQ_ACTIVE_INST_VALU = the launch's SIMD cycles: the VALUs never idle" is withdrawn and these class prices are back.
ISSUE_CYCLES = {"oct": 3.32, "quad": 3.45, "wide": 3.10, "wide/split": 3.22, "quad/split": 3.37}


def lookup_traffic(voices, nframes, kind, form, avg_launch_s):
    """HBM bytes and VALU instructions per launch come from separate rocprofv3 --pmc passes (they cannot be combined
    with the timed run).  profiles/traffic_r04.json holds one entry per (workload, kernel form) that was profiled
    (tools/profile_workload.sh + tools/make_traffic.py), the whole file stamped with kernel_source_hash(): an entry is
    only quoted for THIS workload in THIS kernel form built from THESE kernel sources; otherwise traffic is null."""
    try:
        tj = json.load(open(TRAFFIC_FILE))
    except (OSError, ValueError):
        return None, None, "no PMC profile (profiles/%s missing)" % os.path.basename(TRAFFIC_FILE)
    if tj.get("kernel_source_sha16") != kernel_source_hash():
        return None, None, "profiles/%s is stale (kernel sources changed since the PMC passes): not quoted" % os.path.basename(TRAFFIC_FILE)
    for e in tj.get("entries", []):
        w = e.get("workload", {})
        if (w.get("voices_per_gpu"), w.get("frames_per_voice"), w.get("kind"), w.get("kernel_form")) != (voices, nframes, kind, form):
            continue
        valu = None
        if e.get("SQ_INSTS_VALU") and e.get("issue_cycles_per_valu"):
            # wave64 VALU instructions of one launch (PMC) x the mix-weighted issue cost of the kernel's instruction classes
            # (ISSUE_CYCLES above) against this run's launch time on 1024 SIMDs at the 2.4 GHz peak clock
            valu = {"insts_per_launch": e["SQ_INSTS_VALU"], "issue_cycles_per_inst": e["issue_cycles_per_valu"],
                    "source": e["source"] + "; " + e.get("issue_cycles_source", ""),
                    "frac_of_issue_slots": e["SQ_INSTS_VALU"] * e["issue_cycles_per_valu"] / (avg_launch_s * 2.4e9 * 1024),
                    "wait_any_over_wave_cycles": e.get("SQ_WAIT_ANY_over_WAVE_CYCLES")}
        return e["traffic_bytes_per_launch"], valu, e["source"]
    return None, None, "no PMC profile for this workload / kernel form in profiles/%s" % os.path.basename(TRAFFIC_FILE)


def resolve_workload(a, world):
    """(configs[] index, voices per GPU, kind) of a run: --config names the per-GPU workload, --voices / --workload
    override its parts; without --config: configs[1] on one GPU, configs[4]'s shard per GPU on several."""
    config = a.config if a.config is not None else (4 if world > 1 else 1)
    voices = a.voices if a.voices is not None else CONFIGS[config][0]
    kind = a.workload if a.workload is not None else CONFIGS[config][1]
    return config, voices, kind


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)       # (a step is ~2.5 ms: half a second of GPU work after the CPU baseline's 3 s)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", type=int, default=None, choices=sorted(CONFIGS),
                    help="per-GPU workload = BASELINE.json configs[K] (4: one GPU's shard of it); default 1 at --gpus 1, 4 at --gpus N")
    ap.add_argument("--voices", type=int, default=None, help="voices per GPU (overrides --config's)")
    ap.add_argument("--seconds", type=float, default=1.0)
    ap.add_argument("--workload", default=None, choices=["static", "timevarying", "ragged"],
                    help="overrides --config's: static (config-2 voices), timevarying (config-3 voices) or ragged (configs[3]'s utterances)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-stream", action="store_true", help="leave the default line's `stream` record out")
    ap.add_argument("--no-end-to-end", action="store_true", help="leave configs[3]'s `end_to_end` record out (profiling passes)")
    ap.add_argument("--split", default="auto", help="time split (include/trm_c_api.h): auto | off | control periods per segment")
    ap.add_argument("--mode", default="batch", choices=["batch", "stream"],
                    help="stream: the metric's second half demonstrated -- --voices N (default 1048576) streamed in 100 ms chunks "
                         "through trm_stream_push_device for --seconds (default 2) of audio, PCM left on the device; a step = one chunk")
    ap.add_argument("--kernel", default="auto", choices=["auto", "wide", "quad", "oct"],
                    help="kernel form (include/trm_c_api.h); auto = the library's choice by batch size")
    return ap.parse_args(argv)


def stream_mode(a, rank, world):
    """`--mode stream`: concurrent real-time voices measured the way the reference means them (TRAcT's producer loop,
    Applications/TRAcT/tube.c:1096-1190 + Controller.m:73-100): N voices per GPU streamed in 100 ms chunks, every chunk waited for
    on its own (tools/realtime_voices.py).  value = output samples/s while streaming; `held` = every chunk beat its audio time."""
    if world != 1:
        sys.exit("bench.py --mode stream runs on one GPU (the streams of different GPUs are independent: multiply)")
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import realtime_voices as rv
    N = a.voices if a.voices is not None else 1048576
    seconds = a.seconds if a.seconds != 1.0 else 2.0
    r = rv.run(N, seconds, 25, False)
    per_chunk = N * 4410
    out = {"metric": "audio samples/s (whole node) + concurrent real-time tube voices", "value": per_chunk / (r["median_ms"] * 1e-3), "unit": "samples/s",
           "n_gpus": 1, "steps": r["chunks"], "warmup": 1, "ms_per_step": r["median_ms"], "higher_is_better": True, "scaling": "weak",
           "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "realtime_voices_streamed": N, "held_real_time": r["held"], "chunk_ms_audio": r["chunk_ms_audio"], "chunk_ms_max": r["max_ms"],
           "chunk_ms_p99": r["p99_ms"], "realtime_factor": r["realtime_factor"],
           "config": {"workload": "%d time-varying voices streamed in 100 ms chunks (25 control frames) for %g s of audio through "
                                  "trm_stream_push_device; frames and fp32 PCM resident on the device" % (N, seconds), "mode": "stream"}}
    print(json.dumps(out), flush=True)


def stream_record(N=1048576, chunks=10, chunk_frames=25):
    """The metric's second half on the default line: N voices streamed in `chunks` chunks of 100 ms (25 control frames) through
    trm_stream_push_device, frames and fp32 PCM resident on the device, every chunk waited for on its own."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import torch
    import realtime_voices as rv
    try:
        r = rv.run(N, chunks * chunk_frames / 250.0, chunk_frames, False)
    except Exception as e:          # (a smaller device: report, do not lose the line)
        return {"voices": N, "error": str(e)[:200]}
    finally:
        torch.cuda.empty_cache()
    return {"voices": N, "chunks": r["chunks"], "chunk_ms_audio": r["chunk_ms_audio"], "chunk_ms_median": r["median_ms"],
            "chunk_ms_max": r["max_ms"], "held_real_time": r["held"], "samples_per_s": N * 4410 / (r["median_ms"] * 1e-3),
            "how": "trm_stream_push_device, 25 control frames per push, frames and fp32 PCM on the device, one synchronisation per chunk "
                   "(tools/realtime_voices.py)"}


def end_to_end_record(g, pd, caller_order, reps=3):
    """configs[3] as SURVEY 8(d) defines "end to end through TRM": the C-ABI host entry -- H2D of the frames, the kernels, the
    converter, per-voice maxima, D2H of the PCM -- on the batch in the CALLER's order (the library sorts by length itself),
    called the way a C caller calls it: one packed [sum n][16] fp32 frame array + offsets in, a kept output buffer out (what
    gnuspeech_amd.TRMBatch does before it gets here -- packing 1024 numpy arrays -- is the Python mirror's cost, timed
    separately as `python_mirror_ms`); fp32 PCM and the containers' int16."""
    import ctypes as C
    import numpy as np
    b = g.TRMBatch(g.TRMInputParameters.from_dict(pd))
    L = g.lib()
    V = len(caller_order)
    nfr = np.array([len(u) for u in caller_order], dtype=np.uint32)
    foff = np.concatenate([[0], np.cumsum(nfr[:-1], dtype=np.uint64)]).astype(np.uint64)
    frames = np.ascontiguousarray(np.concatenate([np.asarray(u, dtype=np.float32) for u in caller_order]))
    nout = np.array([b.samples_for_frames(int(n)) for n in nfr], dtype=np.uint64)
    ooff = np.concatenate([[0], np.cumsum(nout[:-1], dtype=np.uint64)]).astype(np.uint64)
    total = int(nout.sum())
    ns, mx = np.zeros(V, dtype=np.uint32), np.zeros(V, dtype=np.float32)
    out = {}
    for name, entry, dtype, extra in (("fp32", L.trm_batch_synthesize_host, np.float32, ()), ("int16", L.trm_batch_synthesize_host_int16, np.int16, (0,))):
        buf = np.ones(total, dtype=dtype)                 # every page touched: the caller keeps this buffer between calls
        ts = []
        for _ in range(reps + 1):
            t0 = time.perf_counter()
            rc = entry(b._h, V, frames.ctypes.data, foff.ctypes.data, nfr.ctypes.data, buf.ctypes.data, ooff.ctypes.data, ns.ctypes.data,
                       mx.ctypes.data, *extra)
            ts.append(time.perf_counter() - t0)
            assert rc == 0 and int(ns.sum()) == total
        out[name] = {"ms": min(ts[1:]) * 1e3, "samples_per_s": total / min(ts[1:]), "output_samples": total,
                     "bytes_over_pcie": int(frames.nbytes + buf.nbytes)}
        buf = None
    t0 = time.perf_counter()
    b.synthesize(caller_order, reuse_output=True)
    t0 = time.perf_counter()
    b.synthesize(caller_order, reuse_output=True)
    out["python_mirror_ms"] = (time.perf_counter() - t0) * 1e3
    out["time_split"] = list(b.last_time_split)
    out["how"] = ("trm_batch_synthesize_host / _host_int16 on a packed frame array (%d voices in the caller's order, %.0f MB of frames in, the PCM "
                  "into a buffer the caller keeps), best of %d; python_mirror_ms = gnuspeech_amd.TRMBatch.synthesize on the list of per-voice "
                  "arrays (packing included)" % (V, frames.nbytes / 1e6, reps))
    return out


def main():
    a = parse_args()

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
    config, voices, workload = resolve_workload(a, world)

    if a.mode == "stream":
        return stream_mode(a, rank, world)

    import numpy as np
    import cases
    pd = cases.monet_default_params(44100.0)
    nframes = int(round(a.seconds * 250)) + 1
    caller_order = None
    # per-rank shard: independent voices, different seed offset per rank (no data-path collective)
    if workload == "ragged":
        caller_order = cases.config4_frames(voices, seed=20250119 + rank)
        frames = sorted(caller_order, key=len, reverse=True)           # resident batch: longest first (gnuspeech_amd/shard.py's order)
        nframes = max(len(u) for u in frames)
        wname = ("configs[3]: GnuTTSServer sentence batch, %d ragged utterances of %.1f - %.1f s (%.0f s of speech, seed 20250119) @ 44.1 kHz, "
                 "resident on the device, longest first" % (voices, min(len(u) for u in frames) / 250.0, nframes / 250.0,
                                                            sum(len(u) - 1 for u in frames) / 250.0))
    elif workload == "static":
        frames = cases.config2_frames(voices, nframes=nframes, seed=20250117 + rank)
        wname = "configs[1]: batch=%d static-vowel tubes x %.3g s @ 44.1 kHz, Monet default voice, fp32" % (voices, a.seconds)
    else:
        frames = cases.config3_frames(voices, nframes=nframes, seed=20250118 + rank)
        wname = "configs[2]: batch=%d time-varying tubes (gnuspeech.input tracks) x %.3g s @ 44.1 kHz" % (voices, a.seconds)
    if dist:
        wname = "configs[4]: %d utterances sharded %d per GPU over %d GPUs (no collective); per GPU = %s" % (voices * world, voices, world, wname)
    elif config == 4 and a.voices is None and a.workload is None:
        wname = "configs[4], ONE GPU's shard: %d of 65536 utterances (what every rank of --gpus 8 runs); = %s" % (voices, wname)

    # the CPU leg first: before this process has a GPU context (rank 0 at N=1 only)
    cpu = None
    if rank == 0 and not dist and not a.no_cpu_baseline:
        cpu = cpu_baseline_ragged(pd, frames) if workload == "ragged" else cpu_baseline(pd, frames)

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
    b.set_time_split(a.split if a.split in ("auto", "off") else int(a.split))
    st = b.prepare_device(frames, device="cuda:%d" % local_rank)
    stream = torch.cuda.current_stream()

    for _ in range(a.warmup):
        b.synthesize_device(st, stream)
    torch.cuda.synchronize()
    solo_ms = None
    if dist:
        # outside the timed region: rank 0 runs its shard ALONE while the others wait -- what a reader needs to turn the
        # timed region below into a weak-scaling efficiency without a second run
        td.barrier()
        if rank == 0:
            ks = max(1, min(a.steps, 50))
            t0 = time.perf_counter()
            for _ in range(ks):
                b.synthesize_device(st, stream)
            torch.cuda.synchronize()
            solo_ms = (time.perf_counter() - t0) / ks * 1e3
    b.kernel_time_ms()                                     # reset the per-launch event accumulator
    if dist:
        td.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        b.synthesize_device(st, stream)
    torch.cuda.synchronize()
    own_dt = time.perf_counter() - t0                      # this rank's own launches + sync, before the closing barrier
    if dist:
        td.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    kern_ms, launches = b.kernel_time_ms()                 # hipEvents on the launch stream
    per_rank_ms = per_rank_kernel_ms = None
    if dist:
        dev = "cpu" if rehearsal else "cuda"
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        td.all_reduce(tmax, op=td.ReduceOp.MAX)
        dt = float(tmax.item())
        # every rank's own numbers, gathered after the timed region: [elapsed incl. the closing barrier, kernel time]
        mine = torch.tensor([own_dt * 1e3 / a.steps, kern_ms / max(1, launches)], dtype=torch.float64, device=dev)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        td.all_gather(allr, mine)
        per_rank_ms = [float(x[0]) for x in allr]
        per_rank_kernel_ms = [float(x[1]) for x in allr]

    samples_per_step_rank = int(st["total_out"])
    total_samples = samples_per_step_rank * a.steps * world
    value = total_samples / dt
    # roofline of the dominant kernel (trm_tube_kernel*): algorithmic bytes per launch =
    # 4 B x output samples + 64 B x frames, / its average launch duration
    frames_total = int(np.asarray(st["nframes_host"]).sum())
    alg_bytes = 4.0 * samples_per_step_rank + 64.0 * frames_total
    avg_launch_s = (kern_ms / max(1, launches)) * 1e-3
    achieved = alg_bytes / avg_launch_s / 1e9
    split = b.last_time_split
    form = b.last_kernel + ("/split" if split[0] else "")
    traffic, valu, traffic_note = lookup_traffic(voices, nframes, workload, form, avg_launch_s)
    kernel_name = {"wide": "trm_tube_kernel<0>", "quad": "trm_tube_kernel_q", "oct": "trm_tube_kernel_o",
                   "wide/split": "trm_tube_kernel<2> (+ trm_phase_period_kernel, trm_phase_segment_kernel)",
                   "quad/split": "trm_tube_kernel_q<true, 2, true> (+ trm_phase_period_kernel, trm_phase_segment_kernel)"}[form]
    out = {
        "metric": "audio samples/s (whole node) + concurrent real-time tube voices",
        "value": value, "unit": "samples/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "realtime_voices_44k1": value / 44100.0,
        "config": {"workload": wname, "voices_per_gpu": voices, "frames_per_voice": nframes,
                   "output_samples_per_voice": samples_per_step_rank // max(1, voices),
                   "tube_rate_hz": b.derived["sampleRate"], "control_rate_hz": 250, "sharding": "voices, no collective",
                   "kernel_form": form,
                   "time_split": ({"segment_periods": split[0], "warmup_periods": split[1]} if split[0] else None)},
        # what binds is VALU issue (a scalar recurrence: ~3.6 wave64 instructions per output sample and lane against 4.36
        # algorithmic bytes); the HBM fraction north_star asks for is `frac` (= `hbm_frac`), priced as the contract says
        "roofline": {"bound": "valu", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "hbm_frac": achieved / HBM_PEAK_GBS,
                     "traffic": traffic, "traffic_source": traffic_note,
                     "kernel": kernel_name, "avg_launch_ms": kern_ms / max(1, launches),
                     "algorithmic_bytes_per_launch": alg_bytes, "valu_issue": valu,
                     "issue_cycles_per_inst_by_class_mix": ISSUE_CYCLES[form],
                     "note": "achieved / peak / frac are the HBM figures BASELINE asks for (algorithmic bytes / device time of a launch, "
                             "hipEvents); the kernel is VALU-issue bound (SURVEY 8d): valu_issue.frac_of_issue_slots = PMC instruction "
                             "count x the class-mix issue cost / (launch time x 1024 SIMDs x 2.4 GHz)"},
        "cpu_baseline": cpu,
    }
    if rank == 0 and not dist and workload == "ragged" and not a.no_end_to_end:
        out["end_to_end"] = end_to_end_record(g, pd, caller_order)
    if rank == 0 and not dist and config == 1 and a.voices is None and a.workload is None and not a.no_stream:
        del st, b
        torch.cuda.empty_cache()
        out["stream"] = stream_record()
    if dist:
        out["per_rank_ms"] = per_rank_ms                   # ms per step, every rank's own clock (launches + device sync)
        out["per_rank_kernel_ms"] = per_rank_kernel_ms     # average tube-kernel launch per rank (hipEvents)
        out["per_gpu_value"] = [samples_per_step_rank / (t * 1e-3) for t in per_rank_ms]
        out["solo_ms_per_step"] = solo_ms                  # rank 0 alone on the same shard, outside the timed region
        out["weak_scaling_efficiency_in_run"] = (solo_ms / (dt / a.steps * 1e3)) if solo_ms else None
        out["scaling_baseline"] = ("python bench.py --gpus 1 %s: the same per-GPU shard on one GPU (--gpus 1 WITHOUT that is "
                                   "configs[1], another workload)" % ("--config %d" % config if (a.voices is None and a.workload is None and a.seconds == 1.0)
                                                                      else "--voices %d --workload %s --seconds %g" % (voices, workload, a.seconds)))
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist:
        td.destroy_process_group()


if __name__ == "__main__":
    main()
