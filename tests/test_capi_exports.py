"""CPU-side checks of the boundary: the C-ABI library loads and exports every symbol
include/trm_c_api.h declares; the text format parser/writer round-trips.  No GPU compute here."""
import os
import re

import numpy as np
import pytest

import cases

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    import gnuspeech_amd
    L = gnuspeech_amd.lib()
    hdr = open(os.path.join(ROOT, "include", "trm_c_api.h")).read()
    declared = set(re.findall(r"\b(trm_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(gnuspeech_amd._capi.EXPORTS)
    for name in declared:
        assert getattr(L, name) is not None
    assert b"gfx950" in L.trm_build_info()


def test_product_does_not_link_the_oracle():
    import subprocess
    import gnuspeech_amd
    out = subprocess.run(["nm", "-D", gnuspeech_amd.LIB_PATH], capture_output=True, text=True).stdout
    assert "trm_oracle" not in out
    for root, _, files in os.walk(os.path.join(ROOT, "gnuspeech_amd")):
        for f in files:
            if f.endswith((".py", ".cc", ".h", ".hip")):
                txt = open(os.path.join(root, f)).read()
                assert not re.search(r"#include[^\n]*oracle|import[^\n]*oracle|oracle_lib|libtrm_oracle|trm_oracle_", txt), f


def test_data_list_reads_reference_sample(tmp_path):
    import gnuspeech_amd as g
    dl = g.TRMDataList.initWithContentsOfFile(cases.GNUSPEECH_INPUT)
    p = dl.inputParameters
    assert len(dl.values) == 344                                   # 343 rows + doubled last (TRMDataList.m:239-241)
    assert (p.outputRate, p.controlRate, p.channels, p.waveform, p.tp, p.tnMin, p.tnMax) == (22050.0, 250.0, 1, 0, 40.0, 16.0, 32.0)
    assert list(p.noseRadius) == [0.0, 1.35, 1.96, 1.91, 1.3, 0.73] and p.usesModulation == 1 and p.mixOffset == 54.0
    assert dl.values[0].valuesString == "-12.000 0.000 0.000 0.000 5.500 2500.000 500.000 0.800 0.890 0.990 0.810 0.760 1.050 1.230 0.010 0.100"
    assert np.array_equal(dl.frame_array()[:-1], cases.load_gnuspeech_rows())
    # writer -> parser round trip (MMSynthesisParameters.m:278-310 + TRMParameters.m:26-43 format)
    out = tmp_path / "rt.trm"
    dl.values = dl.values[:-1]
    dl.writeToFile(out)
    dl2 = g.TRMDataList.initWithContentsOfFile(out)
    assert len(dl2.values) == 344 and np.array_equal(dl2.frame_array()[:-1], cases.load_gnuspeech_rows())
    assert bytes(dl2.inputParameters.c) == bytes(p.c)
    assert g.TRMDataList.initWithContentsOfFile(tmp_path / "missing.trm") is None
    short = tmp_path / "short.trm"
    short.write_text("0\n22050\n250\n")
    assert g.TRMDataList.initWithContentsOfFile(short) is None     # truncated header -> nil


def test_product_fails_loudly_without_gpu():
    """On a box without a GPU every synthesis entry point errors out; nothing falls back to the CPU."""
    import pytest
    import gnuspeech_amd as g
    if g.lib().trm_device_count() > 0:
        pytest.skip("GPU present")
    dl = g.TRMDataList()
    dl.inputParameters = g.TRMInputParameters.from_dict(cases.monet_default_params())
    with pytest.raises(g.TrmError) as ei:
        g.TRMTubeModel.initWithInputData(dl)
    assert ei.value.code == 6                                      # TRM_ENODEVICE


def test_sound_file_writer_against_oracle_scaling(tmp_path):
    """trm_write_sound_file (-saveOutputToFile:error:, TRMTubeModel.m:365-490) is host-side container code: AU / AIFF /
    WAVE headers and the int16 payload must be what the oracle's restatement of the scaling produces (mono and stereo)."""
    import ctypes as C
    import gnuspeech_amd as g
    import oracle_lib as O
    L = g.lib()
    rng = np.random.default_rng(3)
    x = (rng.standard_normal(1000) * 0.01).astype(np.float32)
    mx = float(np.abs(x).max())
    for channels, balance in ((1, 0.0), (2, -0.25)):
        for fmt, (magic, hdr, dt) in {0: (b".snd", 24, ">i2"), 1: (b"FORM", 54, ">i2"), 2: (b"RIFF", 44, "<i2")}.items():
            pd = dict(cases.monet_default_params(22050.0), outputFileFormat=fmt, channels=channels, balance=balance, volume=57.0)
            ip = g.TRMInputParameters.from_dict(pd)
            path = str(tmp_path / ("x%d_%d" % (channels, fmt)))
            assert L.trm_write_sound_file(C.byref(ip.c), x.ctypes.data, x.size, mx, path.encode()) == 0
            raw = open(path, "rb").read()
            assert raw[:4] == magic and len(raw) == hdr + 2 * channels * x.size
            body = np.frombuffer(raw[hdr:], dtype=dt).astype(np.int32)
            ref = O.scale_int16(O.InputParams.from_dict(pd), x.astype(np.float64), mx).astype(np.int32)
            diff = ((body - ref + 32768) % 65536) - 32768          # the file path's x2 stereo gain wraps like the reference's cast
            assert np.max(np.abs(diff)) == 0
            # the container as independent parsers see it (python's sunau / aifc / wave)
            import aifc
            import sunau
            import wave
            rd = {0: sunau.open, 1: aifc.open, 2: wave.open}[fmt](path, "rb")
            assert (rd.getnchannels(), rd.getsampwidth(), rd.getframerate(), rd.getnframes()) == (channels, 2, 22050, x.size)
            payload = rd.readframes(x.size)
            rd.close()
            got = np.frombuffer(payload, dtype="<i2" if fmt == 2 else ">i2").astype(np.int32)
            assert np.array_equal(got, body)


def test_event_frame_count_needs_no_gpu():
    import ctypes as C
    import gnuspeech_amd as g
    s = g._capi.TrmIntonation()
    times = np.array([0, 40, 100], dtype=np.uint32)
    n = C.c_size_t()
    assert g.lib().trm_events_count_frames(times.ctypes.data, 3, C.byref(s), C.byref(n)) == 0
    assert n.value == 25                                           # t = 0, 4, ..., 96


def test_shard_voices():
    """trm_shard_voices: contiguous, covering, balanced by frame count; host-only."""
    import gnuspeech_amd as g
    assert g.shard_voices([10] * 8, 2) == [0, 4, 8]
    assert g.shard_voices([100, 1, 1, 1, 1, 1, 1, 1], 2) == [0, 1, 8]           # one long utterance fills a shard
    assert g.shard_voices([], 3) == [0, 0, 0, 0]
    rng = np.random.default_rng(5)
    for trial in range(20):
        n = rng.integers(150, 1500, size=int(rng.integers(1, 400))).astype(np.uint32)
        G = int(rng.integers(1, 9))
        b = g.shard_voices(n, G)
        assert b[0] == 0 and b[-1] == len(n) and all(x <= y for x, y in zip(b, b[1:]))
        cost = [int(n[lo:hi].sum()) for lo, hi in zip(b, b[1:])]
        assert max(cost) <= n.sum() / G + n.max()                              # within one voice of the even share


def test_pulse_shape_beyond_the_table_is_refused():
    """tp + tnMax above 100 % makes the reference write past its 512-entry wavetable (TRMWavetable.m:86-96, no check
    there): the library (trm_derive, no device needed) and the oracle both refuse."""
    import ctypes as C
    import cases
    import gnuspeech_amd as g
    import oracle_lib as O
    from gnuspeech_amd import shard
    pd = cases.monet_default_params(44100.0)
    pd.update(tp=60.0, tnMin=10.0, tnMax=45.0)
    with pytest.raises(g.TrmError) as e:
        shard.derive(g.TRMInputParameters.from_dict(pd))
    assert e.value.code == 10
    with pytest.raises(RuntimeError):
        O.synthesize(O.InputParams.from_dict(pd), np.zeros((3, 16)))
    pd.update(tp=40.0, tnMin=16.0, tnMax=32.0)                      # the Monet default shape is fine
    shard.derive(g.TRMInputParameters.from_dict(pd))


def test_bench_refuses_a_world_size_that_is_not_gpus(tmp_path):
    """bench.py --gpus N must run N ranks or say so: with WORLD_SIZE set to something else it exits with a message instead of
    reporting n_gpus from the environment (--gpus was parsed and never used)."""
    import subprocess, sys
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, env=env, timeout=120)
    assert r.returncode != 0 and "WORLD_SIZE=1" in (r.stderr + r.stdout)


def test_bench_traffic_file_is_stamped_with_the_kernel_sources(tmp_path, monkeypatch):
    """profiles/traffic_r04.json (what bench.py quotes roofline.traffic / valu_issue from) is a LIST of PMC results keyed by
    workload and kernel form, stamped as a whole with the hash of the kernel sources it was measured on: a stale file is
    refused, an entry is quoted only for its own workload in its own kernel form."""
    import json, sys
    sys.path.insert(0, ROOT)
    import bench
    h = bench.kernel_source_hash()
    assert len(h) == 16 and int(h, 16) >= 0
    entry = {"workload": {"voices_per_gpu": 65536, "frames_per_voice": 251, "kind": "static", "kernel_form": "wide"},
             "traffic_bytes_per_launch": 123, "SQ_INSTS_VALU": 1.0e9, "issue_cycles_per_valu": 3.0, "source": "x", "issue_cycles_source": "y"}
    f = tmp_path / "traffic.json"
    monkeypatch.setattr(bench, "TRAFFIC_FILE", str(f))
    assert bench.lookup_traffic(65536, 251, "static", "wide", 1e-3)[0] is None            # no file
    f.write_text(json.dumps({"kernel_source_sha16": "0" * 16, "entries": [entry]}))
    t, v, note = bench.lookup_traffic(65536, 251, "static", "wide", 1e-3)
    assert t is None and v is None and "stale" in note
    f.write_text(json.dumps({"kernel_source_sha16": h, "entries": [entry]}))
    t, v, note = bench.lookup_traffic(65536, 251, "static", "wide", 1e-3)
    assert t == 123 and abs(v["frac_of_issue_slots"] - 1.0e9 * 3.0 / (1e-3 * 2.4e9 * 1024)) < 1e-12
    assert bench.lookup_traffic(65536, 251, "static", "oct", 1e-3)[0] is None             # another kernel form
    assert bench.lookup_traffic(4096, 251, "static", "wide", 1e-3)[0] is None             # another batch
    # the committed file, where there is one, has this shape
    committed = os.path.join(ROOT, "profiles", "traffic_r04.json")
    if os.path.exists(committed):
        tj = json.load(open(committed))
        assert set(("kernel_source_sha16", "entries")) <= set(tj) and len(tj["entries"]) >= 1
        for e in tj["entries"]:
            assert set(("workload", "traffic_bytes_per_launch", "SQ_INSTS_VALU", "issue_cycles_per_valu", "source")) <= set(e)
            assert e["workload"]["kernel_form"] in ("oct", "quad", "wide", "wide/split")


def test_bench_config_flag_names_the_per_gpu_workload():
    """--config K = BASELINE.json configs[K] per GPU whatever --gpus says: `--gpus 1 --config 4` is exactly one GPU's shard of
    configs[4], the run an N-rank line's `scaling_baseline` names (N = 1 and N > 1 ran different workloads
    with no way to run the same one on one GPU)."""
    import sys
    sys.path.insert(0, ROOT)
    import bench
    assert bench.resolve_workload(bench.parse_args([]), 1) == (1, 4096, "static")
    assert bench.resolve_workload(bench.parse_args(["--gpus", "8"]), 8) == (4, 8192, "timevarying")
    assert bench.resolve_workload(bench.parse_args(["--gpus", "1", "--config", "4"]), 1) == (4, 8192, "timevarying")
    assert bench.resolve_workload(bench.parse_args(["--gpus", "2", "--config", "1"]), 2) == (1, 4096, "static")
    assert bench.resolve_workload(bench.parse_args(["--config", "2"]), 1) == (2, 4096, "timevarying")
    assert bench.resolve_workload(bench.parse_args(["--config", "3"]), 1) == (3, 1024, "ragged")      # the GnuTTSServer sentence batch
    assert bench.parse_args([]).split == "auto" and not bench.parse_args([]).no_stream
    assert bench.resolve_workload(bench.parse_args(["--voices", "65536", "--kernel", "wide"]), 1) == (1, 65536, "static")
    assert bench.parse_args([]).steps == 200          # half a second of GPU work: the driver's sampler sees the run
    assert bench.parse_args([]).mode == "batch" and bench.parse_args(["--mode", "stream", "--voices", "2097152"]).mode == "stream"


def test_no_kernel_spills_or_outgrows_its_register_budget(tmp_path):
    """The code objects inside the built libtrm_hip.so: no kernel has a private (scratch) segment or spilled registers -- a
    228-byte spill in the streaming instance of the one-voice-per-lane kernel cost 13 % of its throughput in round 3 (a kernel
    with scratch pays for it on every dispatch) -- and the pipeline kernels stay within 128 VGPRs: four waves per SIMD is what
    puts two workgroups on a CU."""
    import re, shutil, subprocess
    objdump, readelf = "/opt/rocm/lib/llvm/bin/llvm-objdump", "/opt/rocm/lib/llvm/bin/llvm-readelf"
    if not (os.path.exists(objdump) and os.path.exists(readelf)):
        pytest.skip("ROCm LLVM tools not installed")
    lib = shutil.copy(os.path.join(ROOT, "gnuspeech_amd", "libtrm_hip.so"), tmp_path / "lib.so")
    subprocess.run([objdump, "--offloading", str(lib)], check=True, capture_output=True, cwd=tmp_path)
    cos = [f for f in os.listdir(tmp_path) if "gfx950" in f]
    assert cos, "no gfx950 code object in libtrm_hip.so"
    kernels = {}
    for f in cos:
        notes = subprocess.run([readelf, "--notes", str(tmp_path / f)], check=True, capture_output=True, text=True).stdout
        for blk in notes.split("- .agpr_count:")[1:]:              # one metadata entry per kernel (its first key)
            name = re.search(r"\.name:\s+(_Z\S+)", blk)
            if not name:
                continue
            get = lambda key: int(re.search(key + r":\s+(\d+)", blk).group(1))
            kernels[name.group(1)] = (get(r"\.private_segment_fixed_size"), get(r"\.sgpr_spill_count"), get(r"\.vgpr_spill_count"), get(r"\.vgpr_count"))
    tube = [k for k in kernels if "trm_tube_kernel" in k]
    assert len(tube) == 8, sorted(kernels)           # wide x3 (one-shot, stream, time-split), quad x4 (+ its time-split instance), oct
    for k, (scratch, sspill, vspill, vgprs) in kernels.items():
        assert scratch == 0 and sspill == 0 and vspill == 0, (k, scratch, sspill, vspill)
    for k in tube:
        assert kernels[k][3] <= 128, (k, kernels[k][3])
