"""GPU parity: libtrm_hip.so (through the C ABI) against the oracle and the reference fixtures.

Tolerance (BASELINE.json north_star): RMS error <= 1e-5 on output normalised by the reference's
maximumSampleValue; numberSamples must match exactly.  The HIP path computes the signal in fp32
(fp64 only at the reference's discontinuities), so agreement is a tolerance, not bit-equality.
"""
import os
import re

import numpy as np
import pytest

import cases
import golden_io
import oracle_lib as O

pytestmark = pytest.mark.gpu

RMS_TOL = 1e-5      # ONE tolerance for every fixture, both kernel forms and the randomized voices (north_star)
ALL_CASES = list(golden_io.CASE_NAMES)      # incl. short_tube_downsample: the converter's down-sampling branch


def nrms(x, ref, mx):
    e = (np.asarray(x, dtype=np.float64) - ref) / mx
    return float(np.sqrt(np.mean(e * e)))


@pytest.fixture(scope="module")
def g():
    import gnuspeech_amd
    gnuspeech_amd.lib()
    assert gnuspeech_amd.lib().trm_device_count() >= 1
    return gnuspeech_amd


@pytest.fixture(params=["wide", "quad", "quad1", "oct"])
def form(request, monkeypatch):
    """Every kernel form must meet the same bar (include/trm_c_api.h): one voice per lane, four lanes per voice, eight lanes
    per voice, and the four-lane form's second instance ("quad1": one block per pipeline step, the one that lets two
    workgroups share a CU and that batches of more than 16 voices x CUs run; TRM_QUAD_CUS=1 makes every batch take it).
    TRM_TUBE_KERNEL steers every launch that is left on TRM_KERNEL_AUTO; both variables are read when a batch object is
    created."""
    monkeypatch.setenv("TRM_TUBE_KERNEL", "quad" if request.param == "quad1" else request.param)
    if request.param == "quad1":
        monkeypatch.setenv("TRM_QUAD_CUS", "1")
    else:
        monkeypatch.delenv("TRM_QUAD_CUS", raising=False)
    return request.param


@pytest.mark.parametrize("name", ALL_CASES)
def test_tube_model_matches_reference_fixture(g, form, name):
    """TRMTubeModel -initWithInputData: / -synthesize on the GPU vs what the reference's C tube produced."""
    gold = golden_io.load(name)
    dl = g.TRMDataList()
    dl.inputParameters = g.TRMInputParameters.from_dict(gold["params_dict"])
    dl.values = [g.TRMParameters(r) for r in gold["frames"]]
    tube = g.TRMTubeModel.initWithInputData(dl)
    assert tube is not None
    d = tube.derived()
    cp, sr, pad, taps, inc, _ = (int(x) for x in gold["derived"])
    assert (d["controlPeriod"], d["sampleRate"], d["padSize"], d["firTaps"], d["timeRegisterIncrement"]) == (cp, sr, pad, taps, inc)
    tube.synthesize()
    assert tube.numberSamples == gold["numberSamples"]
    mx = gold["maximumSampleValue"]
    out = tube.samples()
    assert np.all(np.isfinite(out))
    tol = RMS_TOL
    err = nrms(out, gold["samples_f32"].astype(np.float64), mx)
    assert err <= tol, "normalised RMS %.3e > %.1e" % (err, tol)
    assert abs(tube.maximumSampleValue - mx) / mx < 2e-4


def test_noise_sequence_bit_exact(g):
    """The chaotic generator (TRMUtility.m:71-85) + one-zero LP (TRMFilters.m:81-86) run on the GPU in
    fp64; the stored fp32 table must equal the oracle's sequence rounded to fp32, bit for bit."""
    b = g.TRMBatch(g.TRMInputParameters.from_dict(cases.monet_default_params()))
    n = 60000
    assert np.array_equal(b.noise_table(n), O.lp_noise(n).astype(np.float32))


def _batch_vs_oracle(g, pd, voices, tol=RMS_TOL):
    ip = g.TRMInputParameters.from_dict(pd)
    b = g.TRMBatch(ip)
    pcm, ns, mx = b.synthesize(voices)
    op = O.InputParams.from_dict(pd)
    worst = 0.0
    for v, fr in enumerate(voices):
        f32 = np.asarray(fr, dtype=np.float32)
        o = O.synthesize(op, f32.astype(np.float64))      # identical (fp32-representable) control inputs
        assert int(ns[v]) == o["numberSamples"], "voice %d" % v
        if o["numberSamples"] == 0:
            continue
        m = o["maximumSampleValue"]
        if m == 0.0:
            assert np.all(pcm[v] == 0.0)
            continue
        e = nrms(pcm[v], o["samples"], m)
        worst = max(worst, e)
        assert e <= tol, "voice %d normalised RMS %.3e" % (v, e)
        assert abs(float(mx[v]) - m) / m < 2e-4
    return worst


@pytest.mark.parametrize("name", ["pulse_no_rise_tp0", "pulse_no_rise_tp0.05", "narrow_band_20hz"])
def test_corner_cases(g, form, name):
    """tests/cases.py corner_cases (a pulse without a rise, tp ~ 0; a 20 Hz frication band) in every kernel form."""
    pd, frames = cases.corner_cases()[name]
    _batch_vs_oracle(g, pd, [frames, frames[:7].copy()])


def test_batch_static_vowels_config2(g, form):
    """BASELINE config 2 shape at a size the oracle finishes in seconds: 96 voices x 0.2 s."""
    fr = cases.config2_frames(96, nframes=51)
    _batch_vs_oracle(g, cases.monet_default_params(44100.0), list(fr))


def test_batch_time_varying_config3(g, form):
    """BASELINE config 3 shape: gnuspeech.input tracks with per-voice time/pitch offsets (frication on)."""
    fr = cases.config3_frames(70, nframes=81)            # 70 voices: one full wave + a partial wave
    _batch_vs_oracle(g, cases.monet_default_params(44100.0), list(fr))


def test_batch_ragged_config4(g, form):
    """BASELINE config 4 shape: ragged utterances in one launch, incl. 0-, 1- and 2-frame voices."""
    voices = cases.config4_frames(20, lo=3, hi=60)
    rows = cases.load_gnuspeech_rows()
    voices += [np.zeros((0, 16)), rows[5:6].copy(), rows[100:102].copy(), rows[0:130].copy()]
    _batch_vs_oracle(g, cases.monet_default_params(22050.0), voices)


def test_long_utterance(g, form):
    """80 s in one voice (20 001 frames, 1.58 M tube samples, 3.5 M outputs) beside two short ones: the 16.16 time
    register, the noise table and the ring positions far from their start; nothing drifts."""
    rows = cases.load_gnuspeech_rows()
    long_voice = np.concatenate([rows] * (20001 // len(rows) + 1))[:20001]
    worst = _batch_vs_oracle(g, cases.monet_default_params(44100.0), [long_voice, rows[:40].copy(), rows[100:131].copy()])
    assert worst <= RMS_TOL


def test_downsampling_batch(g, form):
    """Tube rate above the output rate (short tubes, 22.05 kHz): TRMSampleRateConverter.m:234-297 on the GPU,
    ragged voices, against the oracle."""
    rows = cases.load_gnuspeech_rows()
    pd = cases.monet_default_params(22050.0)
    pd["length"] = 12.5                                                 # "LgChild" voice, Other/voices.config
    voices = [rows[100:130].copy(), rows[10:70].copy(), rows[200:203].copy(), np.zeros((0, 16)), rows[0:1].copy()]
    voices += [rows[i:i + 20].copy() for i in range(0, 300, 5)]          # 60 more: spans two workgroups
    _batch_vs_oracle(g, pd, voices)


@pytest.mark.parametrize("rate", [16000.0, 8000.0, 11025.0])
def test_downsampling_speech_rates(g, form, rate, monkeypatch):
    """The Monet default tube (19 750 Hz) into 16 / 8 / 11.025 kHz output: the tiled down-sampling kernel (per-phase
    coefficient rows in LDS) vs the oracle, and bit for bit vs the generic kernel that walks the fine table like the
    reference's loops (TRMSampleRateConverter.m:234-297).  Ragged batch incl. 0-, 1-, 2-frame voices."""
    pd = cases.monet_default_params(rate)
    rows = cases.load_gnuspeech_rows()
    voices = cases.config4_frames(11, lo=3, hi=70) + [np.zeros((0, 16)), rows[5:6].copy(), rows[100:102].copy(), rows[0:130].copy()]
    worst = _batch_vs_oracle(g, pd, voices)
    b = g.TRMBatch(g.TRMInputParameters.from_dict(pd))
    pcm, ns, mx = b.synthesize(voices)
    monkeypatch.setenv("TRM_DOWNSAMPLE_GENERIC", "1")
    pcm2, ns2, mx2 = b.synthesize(voices)
    assert np.array_equal(ns, ns2) and np.array_equal(mx, mx2)
    for a, c in zip(pcm, pcm2):
        assert np.array_equal(a, c)


def test_downsampling_reference_extra_lap(g, form, monkeypatch):
    """The reference's down-sampling converter can end an utterance with one more lap of its ring (the last dataEmpty
    finds its end pointer behind the read position and takes it for wrapped, TRMSampleRateConverter.m:160-163 after
    TRMRingBuffer.m:85-93): ~600 further outputs computed from what the ring still holds.  Part of the reference's
    result, so part of ours: 1137 frames at 11.025 kHz hit it, 1136 do not."""
    from gnuspeech_amd import shard
    pd = cases.monet_default_params(11025.0)
    ip = g.TRMInputParameters.from_dict(pd)
    d = shard.derive(ip)
    plain = lambda n: (((n - 1) * d["controlPeriod"] + 2 * d["padSize"]) * 65536 + d["timeRegisterIncrement"] - 1) // d["timeRegisterIncrement"]
    assert shard.samples_for_frames(ip, 1137) == plain(1137) + 572 and shard.samples_for_frames(ip, 1136) == plain(1136)
    rows = cases.load_gnuspeech_rows()
    long_rows = np.concatenate([rows] * 4)
    voices = [long_rows[:1137].copy(), long_rows[100:1236].copy(), long_rows[7:1144].copy()]
    _batch_vs_oracle(g, pd, voices)                                   # counts exact, RMS within tolerance, incl. the extra lap
    b = g.TRMBatch(ip)
    pcm, ns, mx = b.synthesize(voices)
    assert [int(x) for x in ns] == [plain(1137) + 572, plain(1136), plain(1137) + 572]
    monkeypatch.setenv("TRM_DOWNSAMPLE_GENERIC", "1")                 # the generic kernel: the same bits
    pcm2, ns2, mx2 = b.synthesize(voices)
    assert np.array_equal(ns, ns2) and np.array_equal(mx, mx2) and all(np.array_equal(a, c) for a, c in zip(pcm, pcm2))


def test_very_high_rate_ratio_runs_the_lane_form(g, form):
    """96 kHz output from a 26 cm tube: 7.2 outputs per tube sample, more than the four-lane form's converter is fed
    for; the library runs the one-voice-per-lane form whatever was asked for -- a one-shot batch and (round 3: the
    one-voice-per-lane form streams too) a stream of any size alike."""
    pd = cases.monet_default_params(96000.0)
    pd["length"] = 26.0
    rows = cases.load_gnuspeech_rows()
    _batch_vs_oracle(g, pd, [rows[:120].copy(), rows[50:343].copy(), rows[5:6].copy()])
    b = g.TRMBatch(g.TRMInputParameters.from_dict(pd))
    b.synthesize([rows[:30].copy()])
    assert b.last_kernel == "wide"
    s = g.TRMStream(g.TRMInputParameters.from_dict(pd), nvoices=2)
    assert s.kernel == "wide"
    fr = np.stack([rows[:40], rows[100:140]]).astype(np.float32)
    parts = [s.push(fr[:, :7])[0], s.push(fr[:, 7:])[0], s.finish()[0]]
    got = np.concatenate(parts, axis=1)
    op = O.InputParams.from_dict(pd)
    for v in range(2):
        o = O.synthesize(op, fr[v].astype(np.float64))
        assert got.shape[1] == o["numberSamples"] and nrms(got[v], o["samples"], o["maximumSampleValue"]) <= RMS_TOL


def test_eight_lane_form_runs_where_it_applies(g, monkeypatch):
    """TRM_KERNEL_OCT (include/trm_c_api.h): AUTO's choice for batches of up to two 8-voice workgroups per CU; a longer
    batch runs the four-lane form instead, and control periods too short for a form's frame staging (below 16 tube
    samples with eight lanes, below 24 with four: the one-shot kernels keep the control frames in LDS a period ahead)
    run the one-voice-per-lane form, whatever was asked for -- with the same result against the oracle."""
    monkeypatch.delenv("TRM_TUBE_KERNEL", raising=False)
    monkeypatch.delenv("TRM_QUAD_CUS", raising=False)
    rows = cases.load_gnuspeech_rows()
    voices = [rows[:90].copy(), rows[40:200].copy(), rows[5:7].copy()]
    pd = cases.monet_default_params(44100.0)
    b = g.TRMBatch(g.TRMInputParameters.from_dict(pd))
    b.set_time_split("off")                             # (whole utterances: AUTO's choice among the three layouts)
    b.synthesize(voices)
    assert b.last_kernel == "oct"                       # AUTO, small batch
    pd["controlRate"] = 1000.0
    pd["length"] = 30.0                                 # control period 12 tube samples
    b = g.TRMBatch(g.TRMInputParameters.from_dict(pd))
    assert b.derived["controlPeriod"] < 16
    b.set_kernel("oct")
    b.synthesize(voices)
    assert b.last_kernel == "wide"
    _batch_vs_oracle(g, pd, voices)
    pd["length"] = 20.0                                 # control period 18: two eight-sample steps fit
    b = g.TRMBatch(g.TRMInputParameters.from_dict(pd))
    assert 16 <= b.derived["controlPeriod"] < 24
    b.set_time_split("off")
    b.synthesize(voices)
    assert b.last_kernel == "oct"
    _batch_vs_oracle(g, pd, voices)
    b.set_kernel("quad")                                # ... but not the four-lane form's three steps
    b.synthesize(voices)
    assert b.last_kernel == "wide"
    pd["length"] = 14.0                                 # control period 25: every form runs
    for form in ("oct", "quad", "wide"):
        b = g.TRMBatch(g.TRMInputParameters.from_dict(pd))
        assert 24 <= b.derived["controlPeriod"] < 32
        b.set_kernel(form)
        pcm, ns, mx = b.synthesize(voices)
        assert b.last_kernel == form
        op = O.InputParams.from_dict(pd)
        for v, fr in enumerate(voices):
            o = O.synthesize(op, np.asarray(fr, dtype=np.float32).astype(np.float64))
            assert int(ns[v]) == o["numberSamples"], (form, v)
            if o["maximumSampleValue"] == 0.0:             # (two frames of silence)
                assert not np.any(pcm[v])
                continue
            assert nrms(pcm[v], o["samples"], o["maximumSampleValue"]) <= RMS_TOL, (form, v)
    monkeypatch.setenv("TRM_QUAD_CUS", "1")             # (read at create: a "device" of one CU holds 16 voices in this form)
    b = g.TRMBatch(g.TRMInputParameters.from_dict(cases.monet_default_params(44100.0)))
    b.synthesize([rows[:30].copy()] * 16)
    assert b.last_kernel == "oct"
    b.synthesize([rows[:30].copy()] * 17)
    assert b.last_kernel == "quad"


def test_extreme_rate_ratios(g, form):
    """Converter ratios at both ends of the up-sampling range: a 30 cm tube (tube rate ~11.7 kHz, ratio 3.8 at
    44.1 kHz: the converter produces ~15 outputs per pipeline step) and a 15.8 cm tube at 22.05 kHz (ratio 1.008)."""
    rows = cases.load_gnuspeech_rows()
    for length, rate in ((30.0, 44100.0), (15.8, 22050.0)):
        pd = cases.monet_default_params(rate)
        pd["length"] = length
        voices = [rows[i:i + 40].copy() for i in range(0, 200, 11)]
        _batch_vs_oracle(g, pd, voices)


def test_short_control_periods(g, form):
    """Control rates far above Monet's 250 Hz: control periods of 10 and 5 tube samples (the pipelines step 2 or 4
    tube samples at a time, so a period boundary falls inside almost every step)."""
    rows = cases.load_gnuspeech_rows()
    for rate in (2000.0, 4000.0):
        pd = cases.monet_default_params(44100.0)
        pd["controlRate"] = rate
        voices = [rows[i:i + 120].copy() for i in range(0, 150, 7)] + [rows[3:5].copy(), np.zeros((0, 16))]
        _batch_vs_oracle(g, pd, voices)


@pytest.mark.parametrize("seed", range(10))
def test_random_voices_and_tracks(g, form, seed):
    """Random utterance-rate parameters inside the ranges the reference's editors allow (Monet's synthesis-parameter
    panel / TRAcT's sliders) with random time-varying tracks: tube lengths from child to giant (both converter
    branches), either waveform, modulation on and off, any control rate from 100 to 1000 Hz."""
    rng = np.random.default_rng(1000 + seed)
    pd = cases.monet_default_params(float(rng.choice([22050.0, 44100.0])))
    pd.update(controlRate=float(rng.choice([100.0, 250.0, 500.0, 1000.0])), waveform=int(rng.integers(0, 2)),
              tp=float(rng.uniform(20, 45)), tnMin=float(rng.uniform(8, 20)), breathiness=float(rng.uniform(0, 10)),
              length=float(rng.uniform(11.0, 24.0)), temperature=float(rng.uniform(25, 40)), lossFactor=float(rng.uniform(0.1, 3.0)),
              apScale=float(rng.uniform(1.5, 5.0)), mouthCoef=float(rng.uniform(2000, 6000)), noseCoef=float(rng.uniform(2000, 6000)),
              noseRadius=[0.0] + [float(x) for x in rng.uniform(0.5, 2.5, 5)], throatCutoff=float(rng.uniform(500, 3000)),
              throatVol=float(rng.uniform(0, 24)), usesModulation=int(rng.integers(0, 2)), mixOffset=float(rng.uniform(30, 60)))
    pd["tnMax"] = pd["tnMin"] + float(rng.uniform(5, 20))
    voices = []
    for _ in range(5):
        n = int(rng.integers(2, 60))
        knots = max(2, n // 8)
        t = np.linspace(0, knots - 1, n)
        def track(lo, hi):
            return np.interp(t, np.arange(knots), rng.uniform(lo, hi, knots))
        fr = np.stack([track(-10, 6), track(0, 60), track(0, 20), track(0, 40), track(0, 7), track(500, 5000), track(200, 2500)]
                      + [track(0.05, 2.5) for _ in range(8)] + [track(0.0, 1.2)], axis=1)
        voices.append(fr)
    _batch_vs_oracle(g, pd, voices)


def test_tract_defaults_and_sine(g, form):
    rows = cases.load_gnuspeech_rows()
    _batch_vs_oracle(g, cases.tract_default_params(), [cases.static_frames(cases.TRACT_VOWEL_FRAME, 21)])
    p = cases.monet_default_params(44100.0)
    p["waveform"] = 1
    p["usesModulation"] = 0
    _batch_vs_oracle(g, p, [rows[100:140].copy(), rows[20:50].copy()])


def test_error_behaviour(g):
    """init -> nil on length <= 0 (TRMTubeModel.m:204-207); no frames -> silent no-op (:274-277)."""
    dl = g.TRMDataList()
    pd = cases.monet_default_params()
    pd["length"] = 0.0
    dl.inputParameters = g.TRMInputParameters.from_dict(pd)
    assert g.TRMTubeModel.initWithInputData(dl) is None
    dl.inputParameters = g.TRMInputParameters.from_dict(cases.monet_default_params())
    t = g.TRMTubeModel.initWithInputData(dl)
    t.synthesize()
    assert t.numberSamples == 0 and t.maximumSampleValue == 0.0
    with pytest.raises(g.TrmError):
        t.generateWAVData()                                # NSParameterAssert(max != 0), :511


def test_synthesizer_facade_and_writers(g, tmp_path):
    """TRMSynthesizer (no doubling of the last frame) + -generateWAVData / -saveOutputToFile vs the oracle writers."""
    gold = golden_io.load("gnuspeech_window_44k")
    pd = dict(gold["params_dict"])
    sp = dict(sampleRate=pd["outputRate"], masterVolume=pd["volume"], outputChannels=1, balance=0.3,
              glottalPulseShape=0, tp=pd["tp"], tnMin=pd["tnMin"], tnMax=pd["tnMax"], breathiness=pd["breathiness"],
              vocalTractLength=pd["length"], temperature=pd["temperature"], lossFactor=pd["lossFactor"],
              apertureScaling=pd["apScale"], mouthCoef=pd["mouthCoef"], noseCoef=pd["noseCoef"],
              n1=pd["noseRadius"][1], n2=pd["noseRadius"][2], n3=pd["noseRadius"][3], n4=pd["noseRadius"][4],
              n5=pd["noseRadius"][5], throatCutoff=pd["throatCutoff"], throatVolume=pd["throatVol"],
              shouldUseNoiseModulation=True, mixOffset=pd["mixOffset"])
    syn = g.TRMSynthesizer()
    syn.setupSynthesisParameters(sp)
    for r in gold["frames"]:
        syn.addParameters(g.TRMParameters(r))
    tube = syn.synthesize()
    assert tube.numberSamples == gold["numberSamples"]      # N frames -> N-1 periods
    wav = syn.lastWAVData
    pd2 = dict(pd, channels=2, balance=0.3)
    op = O.InputParams.from_dict(pd2)
    o = O.synthesize(op, gold["frames"].astype(np.float32).astype(np.float64))
    ref_wav = O.wav_data(op, o["samples"], o["maximumSampleValue"])
    assert len(wav) == len(ref_wav) and wav[:46] == ref_wav[:46]                # header bytes identical
    a = np.frombuffer(wav[46:], dtype="<i2").astype(np.int32)
    r = np.frombuffer(ref_wav[46:], dtype="<i2").astype(np.int32)
    # the fp32 path's error here is 7e-6 of the maximum at its worst sample (RMS 1e-6) and 1e-6 on the maximum itself:
    # 0.7 LSB after the scale by 32767 x 0.65 -- one LSB where the rounding falls the other way (tools/int16_headroom.py)
    assert np.max(np.abs(a - r)) <= 1 and np.mean(np.abs(a - r)) < 0.05
    for fmt, ext in ((0, "au"), (1, "aiff"), (2, "wav")):
        syn.shouldSaveToSoundFile = True
        syn.fileType = fmt
        syn.filename = str(tmp_path / ("out." + ext))
        syn.synthesize()
        raw = open(syn.filename, "rb").read()
        ref = O.scale_int16(op, o["samples"], o["maximumSampleValue"]).astype(np.int32)
        if fmt == 0:
            assert raw[:4] == b".snd" and len(raw) == 24 + 4 * o["numberSamples"]
            body = np.frombuffer(raw[24:], dtype=">i2").astype(np.int32)
        elif fmt == 1:
            assert raw[:4] == b"FORM" and raw[8:12] == b"AIFF"
            body = np.frombuffer(raw[54:], dtype=">i2").astype(np.int32)
        else:
            assert raw[:4] == b"RIFF" and raw[8:12] == b"WAVE"
            body = np.frombuffer(raw[44:], dtype="<i2").astype(np.int32)
        # balance 0.3 in the file path drives the right channel past full scale (scale*2, TRMTubeModel.m:382-383):
        # the reference's int16 cast wraps there, so compare modulo 2^16
        diff = ((body - ref + 32768) % 65536) - 32768
        assert len(body) == len(ref) and np.max(np.abs(diff)) <= 1      # (x 1.3 in the file path: 0.84 LSB predicted, 1 measured)


def cli_argv(tool):
    """The two builds of the softwareTRM command line tool (Frameworks/Tube/main.m:12-67): "c" = tools/softwaretrm.c, plain C
    over the C ABI, built next to the library by gnuspeech_amd/csrc/Makefile; "py" = tools/softwaretrm.py over ctypes."""
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if tool == "c":
        exe = os.path.join(root, "gnuspeech_amd", "softwaretrm")
        assert os.path.exists(exe), "gnuspeech_amd/softwaretrm missing: make -C gnuspeech_amd/csrc"
        return [exe]
    return [sys.executable, os.path.join(root, "tools", "softwaretrm.py")]


@pytest.mark.parametrize("tool", ["c", "py"])
def test_cli_batch_directory_equals_single_file_mode(g, tmp_path, tool):
    """softwaretrm --batch: every file of a directory in one launch per parameter set; each output file
    is byte-identical to what the reference-shaped single-file mode (Frameworks/Tube/main.m:12-67) writes."""
    import os
    import subprocess
    indir, out_b, out_s = tmp_path / "in", tmp_path / "batch", tmp_path / "single"
    os.makedirs(indir)
    os.makedirs(out_s)
    rows = cases.load_gnuspeech_rows()
    names = []
    for i, (lo, hi, fmt) in enumerate(((0, 40, 2), (50, 75, 2), (100, 160, 2), (10, 30, 0))):
        dl = g.TRMDataList()
        pd = cases.monet_default_params(22050.0)
        pd["outputFileFormat"] = fmt
        dl.inputParameters = g.TRMInputParameters.from_dict(pd)
        dl.values = [g.TRMParameters(r) for r in rows[lo:hi]]
        name = "utt%d.trm" % i
        dl.writeToFile(str(indir / name))
        names.append((name, ".wav" if fmt == 2 else ".au"))
    r = subprocess.run(cli_argv(tool) + ["--batch", str(indir), str(out_b)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert "4 files in 2 launches" in r.stdout
    for name, ext in names:
        single = str(out_s / (name[:-4] + ext))
        r = subprocess.run(cli_argv(tool) + [str(indir / name), single], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        assert open(single, "rb").read() == open(str(out_b / (name[:-4] + ext)), "rb").read()


@pytest.mark.parametrize("fmt,channels", [(0, 1), (1, 2), (2, 2), (2, 1)])
def test_c_cli_writes_the_same_files_as_the_python_cli(g, tmp_path, fmt, channels):
    """tools/softwaretrm.c (the first C caller of the trm_tube_* half of the ABI: trm_data_list_read_file -> trm_tube_create ->
    trm_tube_synthesize -> trm_tube_save_output_to_file) against tools/softwaretrm.py: AU / AIFF / WAVE, mono and stereo,
    byte for byte; and the int16 payload against the ORACLE's scaling of the oracle's samples (<= 1 LSB)."""
    import subprocess
    pd = cases.monet_default_params(22050.0)
    pd.update(outputFileFormat=fmt, channels=channels, balance=0.3, volume=54.0)
    rows = cases.load_gnuspeech_rows()[40:90]
    dl = g.TRMDataList()
    dl.inputParameters = g.TRMInputParameters.from_dict(pd)
    dl.values = [g.TRMParameters(r) for r in rows]
    inp = str(tmp_path / "a.trm")
    dl.writeToFile(inp)
    outs = {}
    for tool in ("c", "py"):
        out = str(tmp_path / ("out_" + tool))
        r = subprocess.run(cli_argv(tool) + [inp, out], capture_output=True, text=True)
        assert r.returncode == 0, (r.stdout, r.stderr)
        outs[tool] = (open(out, "rb").read(), r.stdout)
    assert outs["c"] == outs["py"]
    raw, said = outs["c"]
    # without -v only what -saveOutputToFile: prints unconditionally (TRMTubeModel.m:372-376)
    m = re.fullmatch(r"\nnumber of samples:\t(\d+)\nmaximum sample value:\t(\d+\.\d{4})\nscale:\t\t\t(\d+\.\d{4})\n", said)
    assert m, repr(said)
    hdr = {0: 24, 1: 54, 2: 44}[fmt]
    body = np.frombuffer(raw[hdr:], dtype="<i2" if fmt == 2 else ">i2").astype(np.int32)
    back = g.TRMDataList.initWithContentsOfFile(inp)
    op = O.InputParams.from_dict(pd)
    o = O.synthesize(op, back.frame_array().astype(np.float64))
    ref = O.scale_int16(op, o["samples"], o["maximumSampleValue"]).astype(np.int32)
    assert body.size == ref.size and int(m.group(1)) * channels == ref.size
    diff = ((body - ref + 32768) % 65536) - 32768            # (balance 0.3 x2 wraps the right channel like the reference's cast)
    assert np.max(np.abs(diff)) <= 1


def test_c_cli_usage_and_failures(g, tmp_path):
    """Messages and exit codes of Frameworks/Tube/main.m:18-42."""
    import subprocess
    exe = cli_argv("c")[0]
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 255 and r.stderr == "Usage:  %s [-v] inputFile outputFile\n" % exe
    r = subprocess.run([exe, str(tmp_path / "missing.trm"), str(tmp_path / "x.au")], capture_output=True, text=True)
    assert r.returncode == 255 and r.stderr.endswith("Aborting...\n")
    bad = tmp_path / "bad.trm"
    bad.write_text("0\n22050\n")                                 # truncated utterance-rate header (TRMDataList.m:53-214)
    r = subprocess.run([exe, str(bad), str(tmp_path / "x.au")], capture_output=True, text=True)
    assert r.returncode == 255 and r.stderr.endswith("Aborting...\n")


@pytest.mark.parametrize("tool", ["c", "py"])
def test_cli_verbose_prints_input_data_like_the_reference(g, tmp_path, tool):
    """softwareTRM -v (Frameworks/Tube/main.m:44-64): -printInputData's text (TRMDataList.m:251-330,
    TRMTubeModel.m:599-602) in the reference's formats, then the progress lines."""
    import subprocess
    pd = cases.monet_default_params(22050.0)
    pd.update(outputFileFormat=2, channels=2, balance=-0.25, volume=57.5)
    rows = cases.load_gnuspeech_rows()[30:33]
    dl = g.TRMDataList()
    dl.inputParameters = g.TRMInputParameters.from_dict(pd)
    dl.values = [g.TRMParameters(r) for r in rows]
    inp, out = str(tmp_path / "a.trm"), str(tmp_path / "a.wav")
    dl.writeToFile(inp)
    r = subprocess.run(cli_argv(tool) + ["-v", inp, out], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    back = g.TRMDataList.initWithContentsOfFile(inp)           # (the file path doubles the last row, TRMDataList.m:239-241)
    p = back.inputParameters
    d = g.TRMTubeModel.initWithInputData(back).derived()
    exp = "input file:\t\t%s\n\n" % inp
    exp += "outputFileFormat:\tWAVE\noutputRate:\t\t%.1f Hz\ncontrolRate:\t\t%.2f Hz\n\n" % (p.outputRate, p.controlRate)
    exp += "volume:\t\t\t%.2f dB\nchannels:\t\t%d\nbalance:\t\t%+1.2f\n\n" % (p.volume, p.channels, p.balance)
    exp += "waveform:\t\tPulse\ntp:\t\t\t%.2f%%\ntnMin:\t\t\t%.2f%%\ntnMax:\t\t\t%.2f%%\nbreathiness:\t\t%.2f%%\n\n" % (p.tp, p.tnMin, p.tnMax, p.breathiness)
    exp += "nominal tube length:\t%.2f cm\ntemperature:\t\t%.2f degrees C\nlossFactor:\t\t%.2f%%\n\n" % (p.length, p.temperature, p.lossFactor)
    exp += "apScale:\t\t%.2f cm\nmouthCoef:\t\t%.1f Hz\nnoseCoef:\t\t%.1f Hz\n\n" % (p.apScale, p.mouthCoef, p.noseCoef)
    exp += "".join("n%d:\t\t\t%.2f cm\n" % (i, p.noseRadius[i]) for i in range(1, 6))
    exp += "\nthroatCutoff:\t\t%.1f Hz\nthroatVol:\t\t%.2f dB\n\nmodulation:\t\t%s\nmixOffset:\t\t%.2f dB\n\n" % (
        p.throatCutoff, p.throatVol, "on" if p.usesModulation else "off", p.mixOffset)
    exp += "\nactual tube length:\t%.4f cm\ninternal sample rate:\t%d Hz\ncontrol period:\t\t%d samples (%.4f seconds)\n\n" % (
        d["actualTubeLength"], d["sampleRate"], d["controlPeriod"], np.float32(d["controlPeriod"]) / np.float32(d["sampleRate"]))
    fr = back.frame_array()
    exp += "\n%d control rate input tables:\n\n" % len(fr)
    exp += "glPitch\tglotVol\taspVol\tfricVol\tfricPos\tfricCF\tfricBW" + "".join("\tr%d" % (i + 1) for i in range(8)) + "\tvelum\n"
    for row in fr:
        exp += "\t".join("%.2f" % x for x in row) + "\n"
    exp += "\n\nCalculating floating point samples...\nStarting synthesis\ndone.\n"
    assert r.stdout.startswith(exp), "\n".join(a + "   |   " + b for a, b in zip(r.stdout.splitlines(), exp.splitlines()) if a != b)
    assert r.stdout.rstrip().endswith("Wrote scaled samples to file:  %s" % out)


def test_device_path_and_int16(g, form):
    """Device-buffer entry (what bench.py times) == host-buffer entry; int16 normalisation on device."""
    import torch
    pd = cases.monet_default_params(44100.0)
    fr = cases.config3_frames(130, nframes=31)
    b = g.TRMBatch(g.TRMInputParameters.from_dict(pd))
    pcm, ns, mx = b.synthesize(list(fr))
    st = b.prepare_device(fr)
    b.synthesize_device(st)
    torch.cuda.synchronize()
    out = st["out"].cpu().numpy()
    assert np.array_equal(st["number_samples"].cpu().numpy().astype(np.uint32), ns)
    for v in (0, 63, 64, 129):
        o0 = int(st["out_offset_host"][v])
        assert np.array_equal(out[o0:o0 + int(ns[v])], pcm[v])                 # same kernel, same bits
    pcm16 = b.scale_to_int16_device(st).cpu().numpy()
    op = O.InputParams.from_dict(pd)
    v = 5
    o0 = int(st["out_offset_host"][v])
    ref16 = O.scale_int16(op, pcm[v].astype(np.float64), float(mx[v]))
    assert np.array_equal(pcm16[o0:o0 + int(ns[v])], ref16)
    t, n = b.kernel_time_ms()
    assert n >= 2 and t > 0.0


def test_host_entry_input_forms_and_kept_buffer(g, form):
    """TRMBatch.synthesize: a [V,N,16] array == the list of its voices; the kept output buffer (reuse_output)
    gives the same bits and is overwritten by the next call."""
    pd = cases.monet_default_params(44100.0)
    fr = np.ascontiguousarray(cases.config3_frames(70, nframes=25), dtype=np.float32)
    b = g.TRMBatch(g.TRMInputParameters.from_dict(pd))
    pcm, ns, mx = b.synthesize(list(fr))
    pcm2, ns2, mx2 = b.synthesize(fr, reuse_output=True)
    assert np.array_equal(ns, ns2) and np.array_equal(mx, mx2)
    for v in range(70):
        assert np.array_equal(pcm[v], pcm2[v])
    keep = pcm2[3].copy()
    pcm3, ns3, _ = b.synthesize(fr[::-1].copy(), reuse_output=True)
    assert np.array_equal(pcm3[66], keep) and pcm3[0].base is pcm2[0].base
    pcm4, ns4, _ = b.synthesize(np.zeros((0, 5, 16), np.float32), reuse_output=True)
    assert pcm4 == [] and len(ns4) == 0


def test_several_devices_from_one_process(g, form):
    """trm_multi_synthesize_host: shards on different batch objects (here: the one GPU listed three
    times, one host thread and stream each) return what one batch returns, voice for voice; interleaved output
    ranges are refused."""
    pd = cases.monet_default_params(44100.0)
    voices = cases.config4_frames(37, lo=3, hi=90) + [np.zeros((0, 16)), cases.load_gnuspeech_rows()[5:6].copy()]
    ip = g.TRMInputParameters.from_dict(pd)
    pcm, ns, mx = g.TRMBatch(ip).synthesize(voices)
    m = g.TRMMultiBatch(ip, [0, 0, 0])
    pcm2, ns2, mx2 = m.synthesize(voices)
    assert np.array_equal(ns, ns2) and np.array_equal(mx, mx2)
    for a, b in zip(pcm, pcm2):
        assert np.array_equal(a, b)
    pcm3, ns3, _ = g.TRMMultiBatch(ip, [0]).synthesize(voices[:1])              # fewer voices than shards, one shard
    assert np.array_equal(pcm3[0], pcm[0])
    pcm4, _, _ = m.synthesize(voices[:2])
    assert np.array_equal(pcm4[1], pcm[1])
    p16, ns16, _ = m.synthesize_int16(voices)                                    # the int16 entry over the shards
    q16, _, _ = g.TRMBatch(ip).synthesize_int16(voices)
    assert np.array_equal(ns16, ns) and all(np.array_equal(a, b) for a, b in zip(p16, q16))
    with pytest.raises(NotImplementedError):                                      # host-buffer entries only
        m.prepare_device(voices)
    # voices laid out in REVERSE order in the output buffer: the shards' spans still do not interleave -> accepted;
    # alternating voices between two halves of the buffer: refused
    import ctypes as C
    from gnuspeech_amd._capi import lib
    nfr = np.array([len(v) for v in voices[:8]], dtype=np.uint32)
    foff = np.concatenate([[0], np.cumsum(nfr[:-1])]).astype(np.uint64)
    frames = np.concatenate([np.asarray(v, np.float32).reshape(-1, 16) for v in voices[:8]])
    nout = np.array([m.samples_for_frames(n) for n in nfr], dtype=np.uint64)
    total = int(nout.sum())
    out = np.zeros(total, np.float32)
    nsb, mxb = np.zeros(8, np.uint32), np.zeros(8, np.float32)
    rev = (total - np.cumsum(nout)).astype(np.uint64)
    args = lambda oo: (m._h, 8, frames.ctypes.data, foff.ctypes.data, nfr.ctypes.data, out.ctypes.data, oo.ctypes.data, nsb.ctypes.data, mxb.ctypes.data)
    assert lib().trm_multi_synthesize_host(*args(rev)) == 0
    for v in range(8):
        assert np.array_equal(out[int(rev[v]):int(rev[v]) + int(nsb[v])], pcm[v])
    half = total // 2 + 64
    inter = np.array([(v // 2) * int(nout.max()) + (v % 2) * half for v in range(8)], dtype=np.uint64)
    big = np.zeros(2 * half + 8 * int(nout.max()), np.float32)
    a2 = (m._h, 8, frames.ctypes.data, foff.ctypes.data, nfr.ctypes.data, big.ctypes.data, inter.ctypes.data, nsb.ctypes.data, mxb.ctypes.data)
    assert lib().trm_multi_synthesize_host(*a2) != 0
    assert b"interleave" in lib().trm_last_error()


def test_device_entry_cuts_voices_to_max_nframes(g, form):
    """trm_batch_synthesize_device sizes its tables from max_nframes; a voice claiming more frames is cut to it
    (same samples as the voice's first max_nframes frames), never run past them."""
    import torch
    pd = cases.monet_default_params(44100.0)
    rows = cases.load_gnuspeech_rows()
    b = g.TRMBatch(g.TRMInputParameters.from_dict(pd))
    st = b.prepare_device([rows[:120].copy(), rows[10:50].copy()])
    assert st["max_nframes"] == 120
    st["max_nframes"] = 60
    b.synthesize_device(st)
    torch.cuda.synchronize()
    pcm, ns, mx = b.synthesize([rows[:60].copy(), rows[10:50].copy()])
    got = st["number_samples"].cpu().numpy()
    assert [int(x) for x in got] == [int(x) for x in ns]
    out = st["out"].cpu().numpy()
    for v in range(2):
        o0 = int(st["out_offset_host"][v])
        assert np.array_equal(out[o0:o0 + int(ns[v])], pcm[v])


def test_garbage_control_values_terminate(g, form):
    """NaN, infinities and absurd magnitudes in the control frames: no index is derived from data, so the launch ends,
    the sample counts are those of the frame counts, and the sane voices beside them are untouched."""
    pd = cases.monet_default_params(44100.0)
    rows = cases.load_gnuspeech_rows()
    rng = np.random.default_rng(99)
    bad = []
    for kind in range(6):
        fr = rows[20:60].copy()
        if kind == 0:
            fr[5:, :] = np.nan
        elif kind == 1:
            fr[7, rng.integers(0, 16, 5)] = np.inf
        elif kind == 2:
            fr[:, :] = rng.standard_normal(fr.shape) * 1e30
        elif kind == 3:
            fr[:, 7:15] = 0.0                                  # every radius 0: 0/0 in the scattering coefficients
        elif kind == 4:
            fr[:, 0] = 1e9                                     # pitch: f0 overflows
            fr[:, 4] = -1e9                                    # frication position far outside the tube
        else:
            fr[:, :] = -np.inf
        bad.append(fr)
    good = [rows[100:140].copy(), rows[0:40].copy()]
    b = g.TRMBatch(g.TRMInputParameters.from_dict(pd))
    ref, nsr, _ = b.synthesize(good)
    pcm, ns, mx = b.synthesize(bad[:3] + [good[0]] + bad[3:] + [good[1]])
    assert all(int(n) == int(nsr[0]) for n in ns)
    assert np.array_equal(pcm[3], ref[0]) and np.array_equal(pcm[7], ref[1])


@pytest.mark.parametrize("channels", [1, 2])
def test_host_entry_int16(g, form, channels):
    """trm_batch_synthesize_host_int16: the containers' int16 PCM straight from the device == the oracle's scaling
    (TRMTubeModel.m:370-389 file form, :515-540 WAV-data form) of the same launch's fp32 PCM, ragged batch, mono and
    stereo with balance."""
    pd = cases.monet_default_params(44100.0)
    pd["channels"] = channels
    pd["balance"] = 0.3
    pd["volume"] = 57.0
    voices = cases.config4_frames(9, lo=3, hi=70) + [np.zeros((0, 16)), cases.load_gnuspeech_rows()[5:6].copy()]
    b = g.TRMBatch(g.TRMInputParameters.from_dict(pd))
    pcm, ns, mx = b.synthesize(voices)
    op = O.InputParams.from_dict(pd)
    for wav in (False, True):
        p16, ns16, mx16 = b.synthesize_int16(voices, for_wav_data=wav, reuse_output=wav)
        assert np.array_equal(ns, ns16) and np.array_equal(mx, mx16)
        for v in range(len(voices)):
            if int(ns[v]) == 0 or float(mx[v]) == 0.0:
                continue
            ref = np.asarray(O.scale_int16(op, pcm[v].astype(np.float64), float(mx[v]), for_wav_data=wav)).reshape(p16[v].shape)
            assert np.array_equal(p16[v], ref), (wav, v)


def test_launch_timing_is_bounded_and_complete(g):
    """A caller that never asks for the kernel time must not pile up events: finished launches are folded into running
    sums; asking later still accounts for every launch."""
    import torch
    b = g.TRMBatch(g.TRMInputParameters.from_dict(cases.monet_default_params(44100.0)))
    st = b.prepare_device(cases.config2_frames(4, nframes=6))
    b.kernel_time_ms()
    for _ in range(300):
        b.synthesize_device(st)
    torch.cuda.synchronize()
    t, n = b.kernel_time_ms()
    assert n == 300 and t > 0.0
    t, n = b.kernel_time_ms()
    assert n == 0 and t == 0.0


def test_concurrent_handles_from_threads(g):
    """Handles are independent (include/trm_c_api.h): six threads, each creating fresh batch objects (the process-wide
    noise sequence and the shared device tables under load, both converter branches) and synthesizing concurrently, get
    the bits of a sequential run."""
    import threading
    rows5 = np.concatenate([cases.load_gnuspeech_rows()] * 5)
    jobs = []
    for i, rate in enumerate([44100.0, 16000.0, 22050.0, 8000.0, 48000.0, 11025.0]):
        pd = cases.monet_default_params(rate)
        pd["length"] = 14.0 + i
        jobs.append((pd, [rows5[s:s + n].copy() for s, n in ((3 * i, 600 + 150 * i), (50, 80), (7, 300))]))

    def work(pd, voices, out, k, reps):
        res = []
        for _ in range(reps):
            b = g.TRMBatch(g.TRMInputParameters.from_dict(pd))
            res.append([p.copy() for p in b.synthesize(voices)[0]])
        out[k] = res

    seq, par = {}, {}
    for k, (pd, v) in enumerate(jobs):
        work(pd, v, seq, k, 1)
    threads = [threading.Thread(target=work, args=(pd, v, par, k, 3)) for k, (pd, v) in enumerate(jobs)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    for k in range(len(jobs)):
        assert len(par[k]) == 3
        for res in par[k]:
            assert all(np.array_equal(a, b) for a, b in zip(res, seq[k][0])), k


@pytest.mark.parametrize("rate", [44100.0, 16000.0])
def test_hip_graph_capture_and_replay(g, form, rate):
    """With launch timing off the device entry is pure stream work: captured once into a HIP graph (torch.cuda.graph) and
    replayed, it reproduces the direct launch bit for bit, also after the input frames changed in place.  16 kHz: the
    down-sampling branch (tube kernel + conversion kernel; its row offsets are uploaded when the batch shape changes, not
    per launch, so the entry stays capturable)."""
    import torch
    pd = cases.monet_default_params(rate)
    b = g.TRMBatch(g.TRMInputParameters.from_dict(pd))
    fr = cases.config3_frames(40, nframes=26)
    st = b.prepare_device(fr)
    b.synthesize_device(st)                                   # direct launch: tables in place, reference result
    torch.cuda.synchronize()
    ref = st["out"].clone()
    b.set_timing(False)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        b.synthesize_device(st, side)                         # warm-up on the capture stream
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        b.synthesize_device(st)
    st["out"].zero_()
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(st["out"], ref)
    fr2 = cases.config3_frames(40, nframes=26, seed=7)        # new control tracks into the same device buffer
    flat = torch.from_numpy(np.ascontiguousarray(fr2.reshape(-1, 16), dtype=np.float32)).to(st["frames"].device)
    st["frames"].copy_(flat)
    graph.replay()
    torch.cuda.synchronize()
    replayed = st["out"].clone()
    b.synthesize_device(st)
    torch.cuda.synchronize()
    assert torch.equal(st["out"], replayed) and not torch.equal(replayed, ref)
    t, n = b.kernel_time_ms()
    b.set_timing(True)


@pytest.mark.parametrize("fmt", [0, 1, 2])
@pytest.mark.parametrize("channels", [1, 2])
def test_sound_files_composed_on_the_device(g, tmp_path, fmt, channels):
    """SURVEY 8f N2 "containers on device": trm_batch_sound_files_device writes every voice's complete file image -- header +
    int16 payload in the container's byte order -- on the GPU; byte for byte what the host writer (trm_write_sound_file =
    -saveOutputToFile:error:, TRMTubeModel.m:365-490) writes for the same samples: AU / AIFF / WAVE, mono / stereo, ragged voices
    incl. one of two frames."""
    import ctypes as C
    import torch
    pd = cases.monet_default_params(44100.0)
    pd.update(outputFileFormat=fmt, channels=channels, balance=0.3, volume=57.0)
    ip = g.TRMInputParameters.from_dict(pd)
    voices = cases.config4_frames(9, lo=5, hi=40) + [cases.load_gnuspeech_rows()[3:5].copy()]
    b = g.TRMBatch(ip)
    st = b.prepare_device(voices)
    b.synthesize_device(st)
    files, foff, sizes = b.sound_files_device(st)
    torch.cuda.synchronize()
    img = files.cpu().numpy()
    out = st["out"].cpu().numpy()
    ns = st["number_samples"].cpu().numpy()
    mx = st["max_sample"].cpu().numpy()
    for v in range(len(voices)):
        n = int(ns[v])
        assert int(sizes[v]) == g.lib().trm_sound_file_size(C.byref(ip.c), n)
        samples = np.ascontiguousarray(out[int(st["out_offset_host"][v]):int(st["out_offset_host"][v]) + n])
        path = str(tmp_path / ("v%d" % v)).encode()
        assert g.lib().trm_write_sound_file(C.byref(ip.c), samples.ctypes.data, n, float(mx[v]), path) == 0
        want = np.fromfile(path.decode(), dtype=np.uint8)
        got = img[int(foff[v]):int(foff[v]) + int(sizes[v])]
        assert got.size == want.size and np.array_equal(got, want), (fmt, channels, v)


def test_full_size_properties(g, form):
    """BASELINE config 2 at full size (4096 voices x 1 s): size-independent properties -- exact sample
    counts, finite output, voices with identical tracks give identical bits wherever they sit in the
    batch, and a sampled subset agrees with the oracle."""
    pd = cases.monet_default_params(44100.0)
    fr = cases.config2_frames(4096, nframes=251)
    fr[4095] = fr[0]
    fr[2049] = fr[0]
    b = g.TRMBatch(g.TRMInputParameters.from_dict(pd))
    import torch
    st = b.prepare_device(fr)
    b.synthesize_device(st)
    torch.cuda.synchronize()
    ns = st["number_samples"].cpu().numpy()
    assert np.all(ns == 44159)
    out = st["out"].cpu().numpy().reshape(4096, st["out_alloc"] // 4096)[:, :44159]      # rows start on 128-byte boundaries
    assert np.all(np.isfinite(out))
    assert np.array_equal(out[0], out[4095]) and np.array_equal(out[0], out[2049])
    mx = st["max_sample"].cpu().numpy()
    assert np.allclose(mx, np.abs(out).max(axis=1), rtol=0, atol=0)
    op = O.InputParams.from_dict(pd)
    for v in (0, 777, 4094):
        o = O.synthesize(op, fr[v].astype(np.float32).astype(np.float64))
        assert nrms(out[v], o["samples"], o["maximumSampleValue"]) <= RMS_TOL



def test_more_than_two_rounds_of_workgroups_go_out_in_slices(g):
    """The north_star regime (131 072 voices per GPU = 1e6 per node) at a short utterance: 131 072 + 100 voices x 0.1 s run the
    one-voice-per-lane kernel as three launches (slices of 1024 workgroups, trm_kernels.hip launch_tube).  Size-independent
    properties across the slice boundaries: exact counts everywhere, finite, the reported maximum is the maximum, identical
    tracks give identical bits in whichever slice they sit, and voices at the boundaries against the oracle."""
    import torch
    pd = cases.monet_default_params(44100.0)
    V, n = 131072 + 100, 26
    base = cases.config3_frames(512, nframes=n).astype(np.float32)
    fr = np.tile(base, (V // 512 + 1, 1, 1))[:V].copy()
    b = g.TRMBatch(g.TRMInputParameters.from_dict(pd))
    st = b.prepare_device(fr)
    b.synthesize_device(st)
    torch.cuda.synchronize()
    assert b.last_kernel == "wide"
    want = b.samples_for_frames(n)
    ns = st["number_samples"].cpu().numpy()
    assert np.all(ns == want)
    out = st["out"].cpu().numpy().reshape(V, st["out_alloc"] // V)[:, :want]
    assert np.all(np.isfinite(out))
    mx = st["max_sample"].cpu().numpy()
    assert np.array_equal(mx, np.abs(out).max(axis=1))
    # voice v's track is base[v % 512]: the same bits in the first, second and third slice (65 536 voices each)
    for v in (0, 5, 511):
        for w in (v + 65536 - 512, v + 65536, v + 131072 - 512, v + 131072 if v + 131072 < V else v + 65536 + 512):
            assert np.array_equal(out[v], out[w]), (v, w)
    op = O.InputParams.from_dict(pd)
    checked = 0
    for v in (65535, 65536, 65537, 65600, 131071, 131072, 131073, 131100, V - 1):
        o = O.synthesize(op, fr[v].astype(np.float64))
        if o["maximumSampleValue"] == 0.0:             # (a track that starts with silence)
            assert not np.any(out[v])
            continue
        assert nrms(out[v], o["samples"], o["maximumSampleValue"]) <= RMS_TOL, v
        checked += 1
    assert checked >= 4


def test_full_size_ragged_batch_config3(g):
    """BASELINE configs[3] at FULL size: 1024 ragged utterances (150 - 1500 frames, seed 20250119) through the host-buffer
    entry, in the caller's (unsorted) order -- the library orders the voices by length internally.  Size-independent
    properties: every voice's exact sample count, finite PCM, the reported maximum is the maximum, identical utterances
    give identical bits wherever they sit; and a sample of voices (the shortest, the longest, some in between) against the
    oracle at the one tolerance."""
    pd = cases.monet_default_params(44100.0)
    utt = cases.config4_frames(1024)
    utt[1000] = utt[3].copy()                                   # the same utterance at two places of the batch
    b = g.TRMBatch(g.TRMInputParameters.from_dict(pd))
    pcm, ns, mx = b.synthesize(utt)
    lens = np.array([len(u) for u in utt])
    assert lens.min() >= 150 and lens.max() <= 1500 and len(set(lens.tolist())) > 500
    want = np.array([b.samples_for_frames(int(n)) for n in lens])
    assert np.array_equal(ns, want)
    for v in range(1024):
        assert np.all(np.isfinite(pcm[v])) and pcm[v].size == want[v]
        assert float(mx[v]) == float(np.abs(pcm[v]).max())
    assert np.array_equal(pcm[3], pcm[1000])
    op = O.InputParams.from_dict(pd)
    order = np.argsort(lens)
    for v in (int(order[0]), int(order[-1]), int(order[511]), 3, 777):
        o = O.synthesize(op, np.asarray(utt[v], dtype=np.float32).astype(np.float64))
        assert o["numberSamples"] == int(ns[v])
        assert nrms(pcm[v], o["samples"], o["maximumSampleValue"]) <= RMS_TOL, v


def test_host_entry_leaves_gaps_between_voices_alone(g, form):
    """include/trm_c_api.h promises writes at out + out_offset[v] only: with padded offsets, what lies between the voices
    (and in front of the first one) in the caller's buffer is not touched, fp32 and int16."""
    import ctypes as C
    pd = cases.monet_default_params(44100.0)
    b = g.TRMBatch(g.TRMInputParameters.from_dict(pd))
    voices = [np.asarray(v, dtype=np.float32) for v in cases.config3_frames(5, nframes=12)]
    voices[2] = voices[2][:7]
    ref, ns_ref, _ = b.synthesize(voices)
    nfr = np.array([len(v) for v in voices], dtype=np.uint32)
    foff = np.concatenate([[0], np.cumsum(nfr[:-1])]).astype(np.uint64)
    frames = np.ascontiguousarray(np.concatenate(voices), dtype=np.float32)
    nout = np.array([b.samples_for_frames(int(n)) for n in nfr], dtype=np.uint64)
    pad = 1000
    ooff = (np.concatenate([[0], np.cumsum(nout[:-1] + pad)]) + pad).astype(np.uint64)
    total = int(ooff[-1] + nout[-1] + pad)
    L = g.lib()
    for dtype, entry, extra in ((np.float32, L.trm_batch_synthesize_host, ()), (np.int16, L.trm_batch_synthesize_host_int16, (0,))):
        out = np.full(total, 12345, dtype=dtype)
        ns = np.zeros(5, dtype=np.uint32)
        mx = np.zeros(5, dtype=np.float32)
        rc = entry(b._h, 5, frames.ctypes.data, foff.ctypes.data, nfr.ctypes.data, out.ctypes.data, ooff.ctypes.data,
                   ns.ctypes.data, mx.ctypes.data, *extra)
        assert rc == 0 and np.array_equal(ns, ns_ref)
        mask = np.ones(total, dtype=bool)
        for v in range(5):
            mask[int(ooff[v]):int(ooff[v] + nout[v])] = False
            if dtype is np.float32:
                assert np.array_equal(out[int(ooff[v]):int(ooff[v] + nout[v])], ref[v])
        assert np.all(out[mask] == 12345)



@pytest.mark.parametrize("split", ["auto", "off"])
def test_full_size_configs4_per_gpu_batch(g, split):
    """BASELINE configs[4]'s per-GPU shard at FULL size: 8192 config-3 voices (time-varying gnuspeech.input tracks) x 1 s, as
    AUTO runs it since round 4 -- cut in time, four segments of 64 voices per workgroup, two workgroups on every CU -- and as
    whole utterances (the four-lane form, two co-resident workgroups per CU).  Size-independent properties (exact counts,
    finite, the reported maximum is the maximum, identical tracks give identical bits wherever they sit) and a sample of
    voices against the oracle."""
    import torch
    pd = cases.monet_default_params(44100.0)
    fr = cases.config3_frames(8192, nframes=251)
    fr[8191] = fr[5]
    fr[4100] = fr[5]
    b = g.TRMBatch(g.TRMInputParameters.from_dict(pd))
    b.set_time_split(split)
    st = b.prepare_device(fr)
    b.synthesize_device(st)
    torch.cuda.synchronize()
    if split == "auto":
        assert b.last_kernel == "wide" and b.last_time_split == (55, 30)        # 85 + 3 x 55 periods: four segments of 128 workgroups
    else:
        assert b.last_kernel == "quad" and b.last_time_split == (0, 0)
    ns = st["number_samples"].cpu().numpy()
    assert np.all(ns == 44159)
    out = st["out"].cpu().numpy().reshape(8192, st["out_alloc"] // 8192)[:, :44159]
    assert np.all(np.isfinite(out))
    assert np.array_equal(out[5], out[8191]) and np.array_equal(out[5], out[4100])
    mx = st["max_sample"].cpu().numpy()
    assert np.array_equal(mx, np.abs(out).max(axis=1))
    op = O.InputParams.from_dict(pd)
    for v in (0, 4097, 8190):
        o = O.synthesize(op, fr[v].astype(np.float32).astype(np.float64))
        assert nrms(out[v], o["samples"], o["maximumSampleValue"]) <= RMS_TOL


@pytest.mark.gpu
def test_graft_entry_smoke_runs():
    """the driver's round-end check: __graft_entry__.smoke() (one small batch through every kernel form and a time-split launch,
    against the oracle) must keep passing as the forms change"""
    import importlib
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if root not in sys.path:
        sys.path.insert(0, root)
    entry = importlib.import_module("__graft_entry__")
    entry.smoke()
