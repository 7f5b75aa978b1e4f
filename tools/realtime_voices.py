#!/usr/bin/env python3
"""How many concurrent real-time 44.1 kHz tube voices does ONE GPU carry -- demonstrated, not inferred from throughput.

The reference's notion of a real-time voice is TRAcT's loop (Applications/TRAcT/tube.c:1096-1190 + Controller.m:73-100): a
producer that keeps synthesizing from the current parameters a little ahead of the audio callback.  Here N voices are
streamed through trm_stream_push_device in chunks of 100 ms (25 control frames at 250 Hz) for `seconds` of audio: control
frames are resident on the device (synthetic config-3 tracks), PCM stays on the device (fp32, one buffer per chunk: a
server mixes / encodes it there; returning it to the host costs 176 KB per voice-second and bounds N by PCIe, which
`--host-int16` measures: int16 at a fixed gain into a page-locked host buffer), every chunk is waited for on its own, and a run "holds" N voices if EVERY chunk took less than
its 100 ms of audio.  Prints the chunk-time distribution per N and the largest N that held.

usage: realtime_voices.py [--seconds 2] [--chunk-frames 25] [--voices N1,N2,...] [--host-int16]"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import cases  # noqa: E402
import gnuspeech_amd as g  # noqa: E402


def run(N, seconds, chunk_frames, host_int16):
    pd = cases.monet_default_params(44100.0)
    dev = torch.device("cuda", 0)
    nchunks = int(round(seconds * 250 / chunk_frames))
    base = cases.config3_frames(min(N, 4096), nframes=nchunks * chunk_frames + 1).astype(np.float32)
    frames = torch.from_numpy(base).to(dev)
    if N > frames.shape[0]:
        frames = frames.repeat((N + frames.shape[0] - 1) // frames.shape[0], 1, 1)[:N].contiguous()
    s = g.TRMStream(g.TRMInputParameters.from_dict(pd), nvoices=N)
    chunk_s = chunk_frames / 250.0
    per = (int(chunk_s * 44100) + 64 + 31) & ~31          # rows on 128-byte boundaries: the kernel stores 128-byte pieces of PCM
    out = torch.empty((N, per), dtype=torch.float32, device=dev)
    mx = torch.empty(N, dtype=torch.float32, device=dev)
    host = torch.empty((N, per), dtype=torch.int16, pin_memory=True) if host_int16 else None
    times = []
    # warm-up: one chunk of a throw-away utterance (the stream's buffers, the noise table, the kernels' code objects)
    s.push_device(frames[:, 0:1].contiguous(), out=out, max_out=mx)
    s.push_device(frames[:, 1:1 + chunk_frames].contiguous(), out=out, max_out=mx)
    s.finish_device(out=out, max_out=mx)
    s.push_device(frames[:, 0:1].contiguous(), out=out, max_out=mx)        # the utterance's starting point (no samples yet)
    at = 1
    torch.cuda.synchronize()
    for c in range(nchunks):
        f = frames[:, at:at + chunk_frames].contiguous()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        _, m = s.push_device(f, out=out, max_out=mx)
        if host_int16:
            # what a player needs: int16 at a fixed gain (no per-utterance normalisation in a live stream), on the host
            pcm16 = (out[:, :m] * 32767.0).clamp_(-32768, 32767).to(torch.int16)
            host[:, :m].copy_(pcm16, non_blocking=True)          # page-locked destination: one DMA
        torch.cuda.synchronize()
        times.append(time.perf_counter() - t0)
        at += chunk_frames
    t = np.array(times) * 1e3
    return {"voices": N, "chunk_ms_audio": chunk_s * 1e3, "chunks": nchunks, "max_ms": float(t.max()), "p99_ms": float(np.percentile(t, 99)),
            "median_ms": float(np.median(t)), "first_ms": float(t[0]), "held": bool(t.max() < chunk_s * 1e3),
            "realtime_factor": float(chunk_s * 1e3 / np.median(t))}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=2.0)
    ap.add_argument("--chunk-frames", type=int, default=25)
    ap.add_argument("--voices", default="65536,262144,524288,786432,1048576,1310720,1572864")
    ap.add_argument("--host-int16", action="store_true")
    a = ap.parse_args()
    best = None
    for N in [int(x) for x in a.voices.split(",")]:
        try:
            r = run(N, a.seconds, a.chunk_frames, a.host_int16)
        except Exception as e:           # out of device memory at some N: report and stop
            print(json.dumps({"voices": N, "error": str(e)[:200]}), flush=True)
            break
        r["pcm"] = "int16 on the host (D2H inside the chunk time)" if a.host_int16 else "fp32 left on the device"
        print(json.dumps(r), flush=True)
        if r["held"]:
            best = r
        torch.cuda.empty_cache()
    print(json.dumps({"largest_N_with_every_chunk_under_its_audio_time": best["voices"] if best else None,
                      "p99_ms_at_that_N": best["p99_ms"] if best else None, "chunk_ms": a.chunk_frames / 250.0 * 1e3}), flush=True)


if __name__ == "__main__":
    main()
