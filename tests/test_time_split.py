"""Time-split launches (include/trm_c_api.h: trm_batch_set_time_split): every utterance cut into segments that run side by
side, each from rest a warm-up ahead of its first control period (gnuspeech_amd/csrc/trm_kernels.hip, kModeSegments).

The split path must meet the SAME bar as whole utterances -- normalised RMS <= 1e-5 against the oracle / the reference's
fixtures, exact numberSamples -- on every up-sampling fixture, on ragged batches, and on the voices that forget slowest
(mouth and velum closed: only the damping factor takes energy out of the tube).  The CPU half checks the host model of the
split arithmetic (tests/_emul) against the oracle, so a regression of the warm-up rule shows without a GPU."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import cases
import golden_io
import oracle_lib as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RMS_TOL = 1e-5
UP_CASES = [n for n in golden_io.CASE_NAMES if n != "short_tube_downsample"]

MV_CLOSED = [-12.0, 60.0, 0.0, 0.0, 5.5, 2500.0, 500.0, 0.8, 0.89, 0.99, 0.81, 0.76, 1.05, 1.23, 0.01, 0.0]   # mouth AND velum closed


def nrms(x, ref, mx):
    e = (np.asarray(x, dtype=np.float64) - ref) / mx
    return float(np.sqrt(np.mean(e * e)))


# ---------------------------------------------------------------- CPU: the host model of the split arithmetic
@pytest.fixture(scope="module")
def emul():
    src = os.path.join(ROOT, "tests", "_emul", "trm_emul.cc")
    lib = os.path.join(ROOT, "tests", "_emul", "libtrm_emul.so")
    csrc = os.path.join(ROOT, "gnuspeech_amd", "csrc")
    deps = [src] + [os.path.join(csrc, f) for f in ("trm_lane.h", "trm_quad.h", "trm_oct.h", "trm_setup.cc", "trm_setup.h")]
    if not os.path.exists(lib) or any(os.path.getmtime(d) > os.path.getmtime(lib) for d in deps):
        subprocess.check_call(["g++", "-O2", "-ffp-contract=off", "-fPIC", "-shared", "-std=c++17", "-o", lib, src,
                               os.path.join(csrc, "trm_setup.cc"), "-lm"])
    E = C.CDLL(lib)
    E.trm_emul_synthesize_split.argtypes = [C.POINTER(O.InputParams), C.POINTER(C.c_float), C.c_size_t, C.POINTER(C.c_float), C.c_size_t,
                                            C.POINTER(C.c_uint32), C.POINTER(C.c_float), C.c_uint32, C.c_uint32]
    return E


def warm_periods(pd, control_period):
    """The library's rule (trm_capi.cc split_warm_samples): the slowest pole to the power W = 1e-5, + 64 samples."""
    import math
    rate = control_period * pd["controlRate"]
    nyq = rate / 2.0
    poles = [1.0 - pd["lossFactor"] / 100.0, abs((nyq - pd["mouthCoef"]) / nyq), abs((nyq - pd["noseCoef"]) / nyq),
             abs(1.0 - 2.0 * pd["throatCutoff"] / rate)]
    w = math.ceil(math.log(1e-5) / math.log(max(poles))) + 64
    return (w + control_period - 1) // control_period


def _emul_split(emul, p, fr, seg, warm):
    fr = np.ascontiguousarray(fr, dtype=np.float32)
    cap = len(fr) * 800 + 2000
    out = np.zeros(cap, dtype=np.float32)
    n, m = C.c_uint32(), C.c_float()
    assert emul.trm_emul_synthesize_split(C.byref(p), fr.ctypes.data_as(C.POINTER(C.c_float)), len(fr), out.ctypes.data_as(C.POINTER(C.c_float)),
                                          cap, C.byref(n), C.byref(m), seg, warm) == 0
    return out[:n.value], m.value


@pytest.mark.parametrize("name", ["gnuspeech_input_22k", "tract_vowel_1s", "frication_sweep", "female_15cm_stereo"])
def test_host_model_of_the_split_meets_the_tolerance_on_fixtures(emul, name):
    g = golden_io.load(name)
    p, fr = g["params"], g["frames"]
    o = O.synthesize(p, np.asarray(fr, dtype=np.float32).astype(np.float64))
    cp = int(o["derived"]["controlPeriod"])
    warm = warm_periods(g["params_dict"], cp)
    for seg in (7, 16):
        y, m = _emul_split(emul, p, fr, seg, warm)
        assert len(y) == o["numberSamples"]
        assert nrms(y, o["samples"], o["maximumSampleValue"]) <= RMS_TOL, (name, seg)


def test_host_model_closed_tract_needs_the_full_warm_up(emul):
    """Mouth and velum closed: nothing but the damping factor (0.995 per sample) takes energy out.  With the library's
    warm-up the split voice is at the unsplit path's own error; with a third of it, it is NOT -- the rule is not slack."""
    pd = cases.monet_default_params(44100.0)
    p = O.InputParams.from_dict(pd)
    fr = cases.static_frames(MV_CLOSED, 401)
    o = O.synthesize(p, np.asarray(fr, dtype=np.float32).astype(np.float64))
    warm = warm_periods(pd, int(o["derived"]["controlPeriod"]))
    assert warm == 30
    whole, _ = _emul_split(emul, p, fr, 1 << 30, 0)
    good, _ = _emul_split(emul, p, fr, 50, warm)
    short, _ = _emul_split(emul, p, fr, 50, warm // 3)
    mx = o["maximumSampleValue"]
    e_whole = nrms(whole, o["samples"], mx)
    assert nrms(good, o["samples"], mx) <= max(1.2 * e_whole, 1e-6)
    assert np.abs(good.astype(np.float64) - whole).max() / mx < 5e-6           # worst single sample
    assert nrms(short, o["samples"], mx) > 5 * e_whole


# ---------------------------------------------------------------- GPU
@pytest.fixture(scope="module")
def g():
    import gnuspeech_amd
    gnuspeech_amd.lib()
    assert gnuspeech_amd.lib().trm_device_count() >= 1
    return gnuspeech_amd


def _batch(g, pd, split):
    b = g.TRMBatch(g.TRMInputParameters.from_dict(pd))
    b.set_time_split(split)
    return b


@pytest.mark.gpu
@pytest.mark.parametrize("name", UP_CASES)
@pytest.mark.parametrize("seg", [5, 9, 40])
@pytest.mark.parametrize("form", ["wide", "quad"])
def test_split_launch_matches_reference_fixture(g, name, seg, form):
    """Every up-sampling fixture, cut every `seg` control periods, against what the reference's C tube produced -- in both
    segment instances: one voice per lane (64 voices x one segment per workgroup) and four lanes per voice (16)."""
    gold = golden_io.load(name)
    b = _batch(g, gold["params_dict"], seg)
    b.set_kernel(form)
    pcm, ns, mx = b.synthesize([gold["frames"], gold["frames"][:seg + 40].copy(), gold["frames"][:2].copy()])
    nper = len(gold["frames"]) - 1
    warm = warm_periods(gold["params_dict"], int(gold["derived"][0]))
    split = nper > seg + warm            # (the first segment is seg + warm periods long: a shorter fixture is one segment, whole)
    assert b.last_time_split == ((seg, warm) if split else (0, 0))
    assert b.last_kernel == form
    assert int(ns[0]) == gold["numberSamples"]
    m = gold["maximumSampleValue"]
    assert nrms(pcm[0], gold["samples_f32"].astype(np.float64), m) <= RMS_TOL
    assert abs(float(mx[0]) - m) / m < 2e-4 and float(mx[0]) == float(np.abs(pcm[0]).max())
    # the two shorter voices of the launch (one ends in the second segment, one before the first ends) against the oracle
    for v in (1, 2):
        fr = np.asarray(gold["frames"][:seg + 40] if v == 1 else gold["frames"][:2], dtype=np.float32)
        o = O.synthesize(gold["params"], fr.astype(np.float64))
        assert int(ns[v]) == o["numberSamples"]
        if o["maximumSampleValue"] > 0:
            assert nrms(pcm[v], o["samples"], o["maximumSampleValue"]) <= RMS_TOL


@pytest.mark.gpu
def test_single_utterance_runs_as_segments_of_the_four_lane_form(g):
    """The reference's own call pattern -- ONE utterance per synthesize (TRMSynthesizer.m:118-136) -- under AUTO: a second of
    speech becomes a dozen segments of one four-lane workgroup each; counts exact, the utterance at the tolerance; a batch of
    64 such voices likewise; 4096 of them take the one-voice-per-lane segments."""
    pd = cases.monet_default_params(44100.0)
    rows = cases.load_gnuspeech_rows()
    fr = np.concatenate([rows, rows])[:251]
    b = _batch(g, pd, "auto")
    op = O.InputParams.from_dict(pd)
    o = O.synthesize(op, np.asarray(fr, dtype=np.float32).astype(np.float64))
    for V in (1, 64):
        pcm, ns, mx = b.synthesize([fr] * V)
        assert b.last_kernel == "quad" and b.last_time_split[0] >= 15 and b.last_time_split[1] == 30
        for v in (0, V - 1):
            assert int(ns[v]) == o["numberSamples"]
            assert nrms(pcm[v], o["samples"], o["maximumSampleValue"]) <= RMS_TOL
            assert float(mx[v]) == float(np.abs(pcm[v]).max())
    st = b.prepare_device(np.repeat(np.asarray(fr, dtype=np.float32)[None], 4096, axis=0))
    b.synthesize_device(st)
    assert b.last_kernel == "wide" and b.last_time_split[0] > 0


@pytest.mark.gpu
@pytest.mark.parametrize("form", ["wide", "quad"])
def test_split_ragged_batch_against_oracle(g, form):
    """BASELINE configs[3]'s shape (ragged utterances, frication and aspiration on) split every 25 control periods: exact
    counts, maxima, every voice at the tolerance; 0-, 1- and 2-frame voices ride along; 150 voices = three blocks of 64."""
    pd = cases.monet_default_params(44100.0)
    voices = cases.config4_frames(146, lo=20, hi=160)
    rows = cases.load_gnuspeech_rows()
    voices += [np.zeros((0, 16)), rows[5:6].copy(), rows[100:102].copy(), cases.static_frames(MV_CLOSED, 140)]
    b = _batch(g, pd, 25)
    b.set_kernel(form)
    pcm, ns, mx = b.synthesize(voices)
    assert b.last_time_split == (25, 30) and b.last_kernel == form
    op = O.InputParams.from_dict(pd)
    worst = 0.0
    for v, fr in enumerate(voices):
        o = O.synthesize(op, np.asarray(fr, dtype=np.float32).astype(np.float64))
        assert int(ns[v]) == o["numberSamples"], v
        if o["numberSamples"] == 0 or o["maximumSampleValue"] == 0.0:
            assert not np.any(pcm[v])
            continue
        e = nrms(pcm[v], o["samples"], o["maximumSampleValue"])
        worst = max(worst, e)
        assert e <= RMS_TOL, (v, e)
        assert float(mx[v]) == float(np.abs(pcm[v]).max())
    print("split ragged batch: worst normalised RMS %.2e" % worst)


@pytest.mark.gpu
@pytest.mark.parametrize("rate", [16000.0, 8000.0, 22050.0])
def test_split_down_sampling_batch(g, rate):
    """Tube rate above the output rate (a 12.5 cm tube at 22.05 kHz; Monet's tube into 16 / 8 kHz): the segments write their
    stretches of the tube-rate rows, the down-sampling kernel converts them.  Ragged voices against the oracle, counts exact
    (incl. the reference's extra converter lap where it occurs)."""
    pd = cases.monet_default_params(rate)
    if rate == 22050.0:
        pd["length"] = 12.5
    rows = cases.load_gnuspeech_rows()
    voices = [rows[i:i + 40 + 3 * i].copy() for i in range(0, 70)] + [np.zeros((0, 16)), rows[9:10].copy()]
    b = _batch(g, pd, 11)
    pcm, ns, mx = b.synthesize(voices)
    assert b.last_time_split[0] == 11
    op = O.InputParams.from_dict(pd)
    for v, fr in enumerate(voices):
        o = O.synthesize(op, np.asarray(fr, dtype=np.float32).astype(np.float64))
        assert int(ns[v]) == o["numberSamples"], v
        if o["numberSamples"] and o["maximumSampleValue"] > 0:
            assert nrms(pcm[v], o["samples"], o["maximumSampleValue"]) <= RMS_TOL, v


@pytest.mark.gpu
def test_split_equals_whole_to_rounding_and_is_deterministic(g):
    """The split launch against the whole-utterance launch of the same kernel form: far inside the tolerance (what the
    warm-up leaves: 1e-6 of the forgotten state), and bit-identical from launch to launch."""
    pd = cases.monet_default_params(44100.0)
    fr = cases.config3_frames(70, nframes=201)
    whole = _batch(g, pd, "off")
    whole.set_kernel("wide")
    a, nsa, mxa = whole.synthesize(fr)
    b = _batch(g, pd, 30)
    p1, ns1, mx1 = b.synthesize(fr)
    p2, ns2, mx2 = b.synthesize(fr)
    assert np.array_equal(nsa, ns1)
    for v in range(70):
        assert np.array_equal(p1[v], p2[v])
        m = float(mxa[v])
        # (worst single sample; a nearly silent voice -- peak below 1e-3 -- is fp32 rounding noise in both launches: absolute floor)
        assert np.abs(p1[v].astype(np.float64) - a[v]).max() < max(1e-5 * m, 2e-8), v
        assert nrms(p1[v], a[v].astype(np.float64), m) < 2e-6
    assert np.array_equal(mx1, mx2)


@pytest.mark.gpu
def test_a_tube_that_never_forgets_is_not_split(g):
    """lossFactor 0: the damping factor is 1, the warm-up has no end.  AUTO runs whole utterances; asking for a split by
    name is refused (TRM_ERANGE)."""
    pd = cases.monet_default_params(44100.0)
    pd["lossFactor"] = 0.0
    fr = cases.config3_frames(2048, nframes=61)
    b = _batch(g, pd, "auto")
    b.synthesize(fr)
    assert b.last_time_split == (0, 0)
    b.set_time_split(20)
    with pytest.raises(g._capi.TrmError) as ei:
        b.synthesize(fr)
    assert ei.value.code == g._capi.TRM_ERANGE


@pytest.mark.gpu
def test_narrow_frication_band_falls_back_to_whole_utterances(g):
    """A frication bandwidth below what the warm-up covers (a 5 Hz band-pass rings for seconds): found by the pre-pass on the
    device, and the launch runs whole utterances -- the result is the whole-utterance kernel's, bit for bit.  The same
    batch with Monet's narrowest legal band (250 Hz) does run split."""
    pd = cases.monet_default_params(44100.0)
    fr = cases.config3_frames(66, nframes=121)
    narrow = fr.copy()
    narrow[40, 30:50, 6] = 5.0                                  # one voice, twenty frames
    whole = _batch(g, pd, "off")
    whole.set_kernel("wide")
    ref, nsr, mxr = whole.synthesize(narrow)
    b = _batch(g, pd, 30)
    pcm, ns, mx = b.synthesize(narrow)
    assert b.last_time_split == (30, 30)                        # (set up as a split launch; the device decided otherwise)
    assert np.array_equal(ns, nsr) and np.array_equal(mx, mxr)
    for v in range(66):
        assert np.array_equal(pcm[v], ref[v]), v
    legal = fr.copy()
    legal[40, 30:50, 6] = 250.0
    ref2, _, _ = whole.synthesize(legal)
    pcm2, _, _ = b.synthesize(legal)
    assert not np.array_equal(pcm2[40], ref2[40])               # (split: equal to rounding, not to the bit)
    assert nrms(pcm2[40], ref2[40].astype(np.float64), float(np.abs(ref2[40]).max())) < 2e-6


@pytest.mark.gpu
def test_auto_splits_the_sentence_batch_and_leaves_named_forms_alone(g):
    """AUTO on configs[3]'s shape at a size where the model must split (256 utterances of up to 1500 frames: the longest
    voice's serial chain against 256 idle CUs); a kernel form set by name runs whole utterances."""
    pd = cases.monet_default_params(44100.0)
    utt = cases.config4_frames(256)
    b = _batch(g, pd, "auto")
    pcm, ns, mx = b.synthesize(utt)
    sp, warm = b.last_time_split
    assert sp > 0 and warm == 30 and b.last_kernel in ("wide", "quad")      # (a segment instance of either form: by predicted time)
    op = O.InputParams.from_dict(pd)
    lens = np.array([len(u) for u in utt])
    order = np.argsort(lens)
    for v in (int(order[0]), int(order[-1]), 100):
        o = O.synthesize(op, np.asarray(utt[v], dtype=np.float32).astype(np.float64))
        assert int(ns[v]) == o["numberSamples"]
        assert nrms(pcm[v], o["samples"], o["maximumSampleValue"]) <= RMS_TOL
    b.set_kernel("oct")
    b.synthesize(utt[:64])
    assert b.last_time_split == (0, 0) and b.last_kernel == "oct"


@pytest.mark.gpu
def test_split_launch_is_capturable(g):
    """The device entry of a split launch is stream work only (two memsets, the pre-pass, two kernel launches of which the
    device runs one): captured into a HIP graph and replayed after the frames changed in place."""
    import torch
    pd = cases.monet_default_params(44100.0)
    fr = np.ascontiguousarray(cases.config3_frames(80, nframes=101), dtype=np.float32)
    b = _batch(g, pd, 20)
    st = b.prepare_device(fr)
    b.synthesize_device(st)                       # tables and buffers in place
    torch.cuda.synchronize()
    want = st["out"].clone()
    b.set_timing(False)
    graph = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        with torch.cuda.graph(graph, stream=s):
            b.synthesize_device(st, stream=s)
    st["out"].zero_()
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(st["out"], want)
    fr2 = fr.copy()
    fr2[:, :, 0] += 1.5
    st["frames"].copy_(torch.from_numpy(fr2.reshape(-1, 16)))
    graph.replay()
    torch.cuda.synchronize()
    b2 = _batch(g, pd, 20)
    ref, _, _ = b2.synthesize(fr2)
    off = st["out_offset_host"]
    got = st["out"].cpu().numpy()
    for v in (0, 41, 79):
        assert np.array_equal(got[int(off[v]):int(off[v]) + len(ref[v])], ref[v])


@pytest.mark.gpu
def test_ragged_batch_on_the_device_is_planned_by_its_real_lengths(g):
    """The device-buffer entry sees the lengths only on the device; with a host copy (trm_batch_hint_frames, which the Python
    mirror passes along) AUTO counts the workgroups that have work -- they are launched first (trm_seg_map_kernel) -- and cuts
    a ragged batch into shorter segments than it would a rectangular one of the longest voice.  Same samples either way (to
    the split's rounding), same counts; an explicit split of more segments than fit the chip at once is still exact."""
    import torch
    pd = cases.monet_default_params(44100.0)
    utt = sorted(cases.config4_frames(1024, seed=20250119), key=len, reverse=True)
    b = _batch(g, pd, "auto")
    st = b.prepare_device(utt)
    b.synthesize_device(st)
    torch.cuda.synchronize()
    hinted = b.last_time_split
    ns1, pcm1 = st["number_samples"].cpu().numpy().copy(), st["out"].cpu().numpy().copy()
    blind = dict(st)
    del blind["nframes_host"]
    b.synthesize_device(blind)
    torch.cuda.synchronize()
    rect = b.last_time_split
    ns2, pcm2 = st["number_samples"].cpu().numpy(), st["out"].cpu().numpy()
    assert 0 < hinted[0] < rect[0] and hinted[1] == rect[1] == 30, (hinted, rect)
    assert np.array_equal(ns1, ns2)
    off = st["out_offset_host"]
    for v in (0, 1, 500, 1023):
        a, c = pcm1[off[v]:off[v] + ns1[v]].astype(np.float64), pcm2[off[v]:off[v] + ns1[v]].astype(np.float64)
        assert nrms(a, c, float(np.abs(c).max())) < 2e-6, v
    # the longest and the shortest voice against the oracle under the hinted plan
    op = O.InputParams.from_dict(pd)
    for v in (0, 1023):
        o = O.synthesize(op, np.asarray(utt[v], dtype=np.float32).astype(np.float64))
        assert int(ns1[v]) == o["numberSamples"]
        assert nrms(pcm1[off[v]:off[v] + ns1[v]], o["samples"], o["maximumSampleValue"]) <= RMS_TOL
