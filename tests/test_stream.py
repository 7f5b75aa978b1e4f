"""Streaming synthesis (include/trm_c_api.h trm_stream_*): HOW an utterance is cut into chunks does
not matter, BIT FOR BIT (tube, oscillator, FIR, band-pass, throat and converter state are carried on the device;
chunk boundaries fall anywhere relative to the kernel's 4-sample steps and to the converter's 32-output blocks), and
the streamed utterance equals the one-shot batch path to rounding (the streaming kernel is a separate compile-time
instance of the same source: the compiler fuses multiply-adds differently in the two, so the last bits differ)."""
import numpy as np
import pytest

import cases

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def g():
    import gnuspeech_amd
    assert gnuspeech_amd.lib().trm_device_count() >= 1
    return gnuspeech_amd


FORM = {"now": "quad"}


@pytest.fixture(autouse=True, params=["quad", "wide"])
def stream_form(request, monkeypatch):
    """Every test runs in both streaming kernel forms: four lanes per voice (streams of fewer voices than fill the chip)
    and one voice per lane (larger ones; here forced onto the tests' few voices by TRM_TUBE_KERNEL, which the library
    reads when the stream is created)."""
    monkeypatch.setenv("TRM_TUBE_KERNEL", request.param)
    FORM["now"] = request.param
    return request.param


def nrms(x, ref, mx):
    if mx == 0.0:                                   # a silent voice: both must be silent
        return 0.0 if not np.any(x) and not np.any(ref) else np.inf
    e = (np.asarray(x, dtype=np.float64) - np.asarray(ref, dtype=np.float64)) / mx
    return float(np.sqrt(np.mean(e * e)))


def stream_all(g, pd, fr, chunks):
    s = g.TRMStream(g.TRMInputParameters.from_dict(pd), nvoices=fr.shape[0])
    assert s.kernel == FORM["now"]
    parts, maxes, pos = [], [], 0
    for c in chunks:
        out, m = s.push(fr[:, pos:pos + c])
        parts.append(out)
        maxes.append(m)
        pos += c
    out, m = s.finish()
    parts.append(out)
    maxes.append(m)
    return np.concatenate(parts, axis=1), np.max(np.stack(maxes), axis=0), s


def one_shot(g, pd, frames):
    b = g.TRMBatch(g.TRMInputParameters.from_dict(pd))
    b.set_kernel(FORM["now"])
    pcm, ns, mx = b.synthesize(list(frames))
    return pcm, ns, mx


@pytest.mark.parametrize("rate,chunks", [(44100.0, [5, 1, 1, 17, 2, 30, 4]), (22050.0, [60]), (44100.0, [1] * 12 + [48]),
                                          (22050.0, [2, 58])])
def test_chunked_equals_one_shot(g, rate, chunks):
    pd = cases.monet_default_params(rate)
    V, n = 21, sum(chunks)
    fr = cases.config3_frames(V, nframes=n).astype(np.float32)
    whole, whole_max, _ = stream_all(g, pd, fr, [n])                 # the utterance in one push
    got, got_max, _ = stream_all(g, pd, fr, chunks)
    assert np.array_equal(got.view(np.uint32), whole.view(np.uint32))
    assert np.array_equal(got_max, whole_max)
    pcm, ns, mx = one_shot(g, pd, fr)                                 # the batch path
    assert got.shape[1] == int(ns[0])
    for v in range(V):
        assert nrms(got[v], pcm[v], mx[v]) <= 2e-6, "voice %d" % v
    assert np.allclose(got_max, mx, rtol=1e-4)


@pytest.mark.parametrize("rate,chunks,length", [(16000.0, [5, 1, 1, 17, 2, 30, 4], None), (8000.0, [60], None), (16000.0, [1] * 12 + [48], None),
                                                 (11025.0, [2, 58], None), (8000.0, [3, 3, 1, 40, 13], None),
                                                 (8000.0, [7, 1, 30, 22], 13.79), (8000.0, [60], 12.2)])
def test_chunked_equals_one_shot_downsampling(g, rate, chunks, length):
    """The same invariants for output rates below the tube rate (19 750 Hz -> 16 / 11.025 / 8 kHz): the chunk's tube
    samples go through HBM behind a history of 2*pad samples and the tiled down-sampling kernel converts the outputs
    whose read position lies in the chunk.  A short tube (13.8 / 12.2 cm: 25.1 / 28.4 kHz -> 8 kHz) is the widest converter a
    stream takes: its tile needs more than 48 KB of LDS (round 4: such a stream was created and then failed at its first push)."""
    pd = cases.monet_default_params(rate)
    if length is not None:
        pd["length"] = length
    V, n = 21, sum(chunks)
    fr = cases.config3_frames(V, nframes=n).astype(np.float32)
    whole, whole_max, _ = stream_all(g, pd, fr, [n])
    got, got_max, _ = stream_all(g, pd, fr, chunks)
    assert np.array_equal(got.view(np.uint32), whole.view(np.uint32))
    assert np.array_equal(got_max, whole_max)
    pcm, ns, mx = one_shot(g, pd, fr)
    assert got.shape[1] == int(ns[0])
    for v in range(V):
        assert nrms(got[v], pcm[v], mx[v]) <= 2e-6, "voice %d" % v
    assert np.allclose(got_max, mx, rtol=1e-4)


def test_second_utterance_starts_from_rest(g):
    pd = cases.monet_default_params(44100.0)
    fr = cases.config3_frames(3, nframes=20).astype(np.float32)
    pcm, ns, mx = one_shot(g, pd, fr)
    s = g.TRMStream(g.TRMInputParameters.from_dict(pd), nvoices=3)
    first = None
    for _ in range(2):
        a, _m = s.push(fr[:, :9])
        b, _m = s.push(fr[:, 9:])
        c, _m = s.finish()
        got = np.concatenate([a, b, c], axis=1)
        if first is None:
            first = got
        assert np.array_equal(got, first)                 # the second utterance starts from rest like the first
        for v in range(3):
            assert nrms(got[v], pcm[v], mx[v]) <= 2e-6


def test_held_parameters_like_tract(g):
    """TRAcT's real-time loop keeps synthesizing from the current parameter set: pushing the same frame every control
    period gives the steady vowel the one-shot path gives for static frames."""
    pd = cases.tract_default_params()
    fr = cases.static_frames(cases.TRACT_VOWEL_FRAME, 40).astype(np.float32)
    pcm, ns, mx = one_shot(g, pd, [fr])
    s = g.TRMStream(g.TRMInputParameters.from_dict(pd), nvoices=1)
    parts = [s.push(fr[i:i + 1])[0] for i in range(40)]
    parts.append(s.finish()[0])
    got = np.concatenate(parts, axis=1)[0]
    assert got.size == int(ns[0]) and nrms(got, pcm[0], mx[0]) <= 2e-6
    assert np.all(np.isfinite(got)) and np.abs(got).max() > 0


@pytest.mark.parametrize("seed", range(6))
def test_random_parameters_and_chunkings(g, seed):
    """Random voices (tube lengths, rates, control rates, waveforms) cut at random places: the cut never shows."""
    rng = np.random.default_rng(700 + seed)
    pd = cases.monet_default_params(float(rng.choice([22050.0, 44100.0])))
    pd.update(controlRate=float(rng.choice([100.0, 250.0, 1000.0])), waveform=int(rng.integers(0, 2)),
              length=float(rng.uniform(16.5, 28.0)), lossFactor=float(rng.uniform(0.1, 3.0)), usesModulation=int(rng.integers(0, 2)),
              breathiness=float(rng.uniform(0, 10)), tp=float(rng.uniform(20, 45)))
    V, n = int(rng.integers(1, 40)), int(rng.integers(8, 90))
    fr = cases.config3_frames(V, nframes=n).astype(np.float32)
    cuts = np.sort(rng.choice(np.arange(1, n), size=min(n - 1, int(rng.integers(1, 9))), replace=False))
    chunks = list(np.diff(np.concatenate([[0], cuts, [n]])))
    whole, whole_max, _ = stream_all(g, pd, fr, [n])
    got, got_max, _ = stream_all(g, pd, fr, chunks)
    assert np.array_equal(got.view(np.uint32), whole.view(np.uint32)), chunks
    assert np.array_equal(got_max, whole_max)
    pcm, ns, mx = one_shot(g, pd, fr)
    assert got.shape[1] == int(ns[0])
    for v in range(V):
        assert nrms(got[v], pcm[v], mx[v]) <= 4e-6, "voice %d" % v


@pytest.mark.parametrize("rate,seed", [(44100.0, 0), (22050.0, 1), (16000.0, 2), (8000.0, 3), (44100.0, 4), (11025.0, 5)])
def test_stream_matches_oracle(g, rate, seed):
    """The streamed utterance against the ORACLE (not against another HIP path): random cuts, both converter branches,
    exact sample counts, normalised RMS <= 1e-5 -- the same bar as the one-shot path (tests/test_gpu_parity.py)."""
    import oracle_lib as O
    rng = np.random.default_rng(900 + seed)
    pd = cases.monet_default_params(rate)
    V, n = 5, 40
    fr = cases.config3_frames(V, nframes=n, seed=20250118 + seed).astype(np.float32)
    cuts = np.sort(rng.choice(np.arange(1, n), size=int(rng.integers(1, 7)), replace=False))
    chunks = [int(c) for c in np.diff(np.concatenate([[0], cuts, [n]]))]
    got, got_max, _ = stream_all(g, pd, fr, chunks)
    op = O.InputParams.from_dict(pd)
    for v in range(V):
        o = O.synthesize(op, fr[v].astype(np.float64))
        assert got.shape[1] == o["numberSamples"], (chunks, got.shape[1], o["numberSamples"])
        e = nrms(got[v], o["samples"], o["maximumSampleValue"])
        assert e <= 1e-5, "voice %d: normalised RMS %.3e, chunks %s" % (v, e, chunks)
        assert abs(float(got_max[v]) - o["maximumSampleValue"]) / o["maximumSampleValue"] < 2e-4


@pytest.mark.parametrize("name", ["tract_mode_ee_step", "tract_mode_fricative", "tract_mode_fric_step"])
def test_stream_against_the_reference_in_tract_order(g, name):
    """The one-voice stream in TRM_STREAM_MODE_TRACT (what TRAcT's real-time loop maps onto), one control period per
    push, against the REFERENCE's tube.c stepped in TRAcT's own loop order (tests/golden/tract_mode_*, oracle/ref_driver.c
    `tract`: parameters held and STEPPED, x10 frication taps, x100): the whole utterance -- the stepped periods and the
    fricative stretches included -- index for index, exact count, at the one tolerance; and every control period on
    its own as well."""
    import golden_io
    gold = golden_io.load(name)
    fr = gold["frames"].astype(np.float32)
    want, mx = gold["samples_f32"].astype(np.float64), gold["maximumSampleValue"]
    s = g.TRMStream(g.TRMInputParameters.from_dict(gold["params_dict"]), nvoices=1, mode="tract")
    parts, peaks = [], []
    for i in range(1, len(fr)):                       # frame f is what control period f - 1 runs on (ref_driver.c:114-121)
        o, m = s.push(fr[i:i + 1])
        parts.append(o[0]); peaks.append(float(m[0]))
    o, m = s.finish()
    parts.append(o[0]); peaks.append(float(m[0]))
    got = np.concatenate(parts).astype(np.float64)
    assert got.size == gold["numberSamples"]
    assert nrms(got, want, mx) <= 1e-5
    per = 441
    assert max(nrms(got[i:i + per], want[i:i + per], mx) for i in range(0, got.size - per, per)) <= 1e-5
    assert abs(max(peaks) - mx) / mx < 2e-4


@pytest.mark.parametrize("rate,seed", [(44100.0, 0), (22050.0, 1), (16000.0, 2), (8000.0, 3)])
def test_tract_mode_stream_matches_oracle(g, rate, seed):
    """TRM_STREAM_MODE_TRACT away from the goldens' one voice: several voices, random time-varying tracks with frication
    and aspiration, random cuts (several periods per push), both converter branches -- against the ORACLE in TRAcT's loop
    order (trm_oracle_synthesize_tract, itself bit-exact against the reference binary in that order,
    tests/test_oracle_golden.py)."""
    import oracle_lib as O
    rng = np.random.default_rng(1900 + seed)
    pd = cases.monet_default_params(rate)
    V, n = 4, 36
    fr = cases.config3_frames(V, nframes=n, seed=20250300 + seed).astype(np.float32)
    cuts = np.sort(rng.choice(np.arange(1, n), size=int(rng.integers(1, 6)), replace=False))
    chunks = [int(c) for c in np.diff(np.concatenate([[0], cuts, [n]]))]
    s = g.TRMStream(g.TRMInputParameters.from_dict(pd), nvoices=V, mode="tract")
    parts, at = [], 0
    for c in chunks:
        parts.append(s.push(fr[:, at:at + c])[0]); at += c
    parts.append(s.finish()[0])
    got = np.concatenate(parts, axis=1)
    op = O.InputParams.from_dict(pd)
    for v in range(V):
        # the oracle ignores frame 0 (ref_driver.c `tract` starts at f = 1): every pushed frame is one period
        o = O.synthesize(op, np.concatenate([fr[v][:1], fr[v]]).astype(np.float64), tract=True)
        assert got.shape[1] == o["numberSamples"], (chunks, got.shape[1], o["numberSamples"])
        e = nrms(got[v], o["samples"], o["maximumSampleValue"])
        assert e <= 1e-5, "voice %d: normalised RMS %.3e, chunks %s" % (v, e, chunks)
    # a mode change in the middle of an utterance is refused
    s2 = g.TRMStream(g.TRMInputParameters.from_dict(pd), nvoices=1)
    s2.push(fr[:1, :2])
    with pytest.raises(Exception):
        s2.set_mode("tract")
    s2.finish()
    s2.set_mode("tract")


@pytest.mark.parametrize("rate,mode", [(44100.0, "framework"), (16000.0, "framework"), (44100.0, "tract")])
def test_device_buffer_entries_equal_the_host_entries(g, rate, mode):
    """trm_stream_push_device / _finish_device (frames and PCM stay on the device, asynchronous, nothing crosses PCIe) return
    the host entries' samples bit for bit -- both converter branches, both loop orders, uneven chunks, into a caller's
    buffer with a pitch of its own; and host and device calls of one stream mix."""
    import torch
    pd = cases.monet_default_params(rate)
    V, n = 19, 30
    fr = cases.config3_frames(V, nframes=n, seed=20250411).astype(np.float32)
    chunks = [1, 7, 2, 20]
    want, want_max, _ = None, None, None
    s = g.TRMStream(g.TRMInputParameters.from_dict(pd), nvoices=V, mode=mode)
    parts, at = [], 0
    for c in chunks:
        parts.append(s.push(fr[:, at:at + c])[0]); at += c
    parts.append(s.finish()[0])
    want = np.concatenate(parts, axis=1)
    d = g.TRMStream(g.TRMInputParameters.from_dict(pd), nvoices=V, mode=mode)
    dev = torch.device("cuda", 0)
    frd = torch.from_numpy(fr).to(dev)
    big = torch.zeros((V, want.shape[1] + 77), dtype=torch.float32, device=dev)       # the caller's own buffer and pitch
    mx = torch.zeros(V, dtype=torch.float32, device=dev)
    got, at, pos = [], 0, 0
    for i, c in enumerate(chunks):
        if i == 2:                                  # a host-buffer call in the middle of device-buffer ones
            o = d.push(fr[:, at:at + c])[0]
            big[:, pos:pos + o.shape[1]] = torch.from_numpy(o).to(dev)
            m = o.shape[1]
        else:
            _, m = d.push_device(frd[:, at:at + c].contiguous(), out=big[:, pos:], max_out=mx)
        at += c
        pos += m
    _, m = d.finish_device(out=big[:, pos:], max_out=mx)
    pos += m
    torch.cuda.synchronize()
    assert pos == want.shape[1]
    assert np.array_equal(big[:, :pos].cpu().numpy().view(np.uint32), want.view(np.uint32))
    assert float(big[:, pos:].abs().max()) == 0.0                                      # nothing written past the samples


def test_chunks_on_different_hip_streams_are_ordered(g):
    """successive calls of one stream may name different HIP streams (host entries run on the object's own,
    device entries on the caller's): every chunk is ordered behind the one before it by an event.  Many voices, chunks
    alternating between two NON-BLOCKING torch streams and the host entry, no synchronisation in between: the samples of the
    all-host sequence, bit for bit."""
    import torch
    pd = cases.monet_default_params(44100.0)
    V, n = 4096, 41
    fr = np.tile(cases.config3_frames(64, nframes=n, seed=7).astype(np.float32), (V // 64, 1, 1))
    chunks = [1, 10, 10, 10, 10]
    s = g.TRMStream(g.TRMInputParameters.from_dict(pd), nvoices=V)
    parts, at = [], 0
    for c in chunks:
        parts.append(s.push(fr[:, at:at + c])[0]); at += c
    parts.append(s.finish()[0])
    want = np.concatenate(parts, axis=1)
    d = g.TRMStream(g.TRMInputParameters.from_dict(pd), nvoices=V)
    dev = torch.device("cuda", 0)
    frd = torch.from_numpy(fr).to(dev)
    big = torch.zeros((V, want.shape[1] + 32), dtype=torch.float32, device=dev)
    sa, sb = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)      # non-blocking: no implicit order with anything
    torch.cuda.synchronize()
    at = pos = 0
    for i, c in enumerate(chunks):
        piece = frd[:, at:at + c].contiguous()
        torch.cuda.synchronize() if i == 0 else None       # (the slices above were made on the default stream)
        with torch.cuda.stream(sa if i % 2 == 0 else sb):
            piece.record_stream(torch.cuda.current_stream())
            _, m = d.push_device(piece, out=big[:, pos:])
        at += c
        pos += m
    with torch.cuda.stream(sa):
        _, m = d.finish_device(out=big[:, pos:])
    pos += m
    torch.cuda.synchronize()
    assert pos == want.shape[1]
    assert np.array_equal(big[:, :pos].cpu().numpy().view(np.uint32), want.view(np.uint32))


def test_tract_order_stream_of_more_voices_than_a_grid_dimension(g):
    """TRAcT order's x100 (trm_gain_kernel) once put the voice index in gridDim.y (limit 65 535); a stream of
    70 000 voices in that order must come back, voices at both ends equal to the same tracks in a small stream."""
    pd = cases.tract_shim_params()
    base = cases.config3_frames(8, nframes=4, seed=3).astype(np.float32)
    V = 70000
    fr = np.tile(base, (V // 8, 1, 1))
    big = g.TRMStream(g.TRMInputParameters.from_dict(pd), nvoices=V, mode="tract")
    small = g.TRMStream(g.TRMInputParameters.from_dict(pd), nvoices=8, mode="tract")
    if big.kernel != small.kernel:          # (same arithmetic either way; same kernel form = same bits)
        pytest.skip("forms differ")
    a = np.concatenate([big.push(fr)[0], big.finish()[0]], axis=1)
    b = np.concatenate([small.push(base)[0], small.finish()[0]], axis=1)
    assert np.all(np.isfinite(a)) and float(np.abs(a).max()) > 0.0
    for v in (0, 7, 65535, 65536, V - 1):
        assert np.array_equal(a[v], b[v % 8]), v


def test_large_wide_stream_across_launch_slices(g, stream_form):
    """A stream of more voices than one launch slice of the one-voice-per-lane kernel holds (65 536): its per-workgroup state
    blocks, the sliced launches and the device-buffer entries together -- identical tracks give identical bits in whichever
    slice they sit, chunked == single push bit for bit, and voices at the slice boundary agree with the oracle."""
    import torch
    import oracle_lib as O
    if stream_form != "wide":
        pytest.skip("the one-voice-per-lane form's launch slices")
    pd = cases.monet_default_params(44100.0)
    V, n = 65536 + 200, 13
    base = cases.config3_frames(256, nframes=n, seed=20250512).astype(np.float32)
    fr = np.tile(base, (V // 256 + 1, 1, 1))[:V].copy()
    dev = torch.device("cuda", 0)
    frd = torch.from_numpy(fr).to(dev)

    def run(cuts):
        s = g.TRMStream(g.TRMInputParameters.from_dict(pd), nvoices=V)
        assert s.kernel == "wide"
        parts, at = [], 0
        for c in cuts:
            o, m = s.push_device(frd[:, at:at + c].contiguous())
            parts.append(o.clone()); at += c
        o, m = s.finish_device(device=dev)
        parts.append(o.clone())
        torch.cuda.synchronize()
        return torch.cat(parts, dim=1).cpu().numpy()
    whole = run([n])
    cut = run([1, 5, 7])
    assert np.array_equal(whole.view(np.uint32), cut.view(np.uint32))
    for v in (0, 3, 199):                               # voice v and v + 65 536 run the same track in different slices
        assert np.array_equal(whole[v], whole[v + 65536]), v
    op = O.InputParams.from_dict(pd)
    checked = 0
    for v in (65535, 65536, 65537, 65600, V - 1):
        o = O.synthesize(op, fr[v].astype(np.float64))
        assert whole.shape[1] == o["numberSamples"]
        if o["maximumSampleValue"] == 0.0:
            continue
        assert nrms(whole[v], o["samples"], o["maximumSampleValue"]) <= 1e-5, v
        checked += 1
    assert checked >= 2


def test_stream_corner_calls(g):
    """finish without a push, a first push of one frame (no samples yet), pushes after finish, device entries with nothing to return."""
    import torch
    pd = cases.monet_default_params(44100.0)
    fr = cases.config3_frames(3, nframes=9, seed=7).astype(np.float32)
    s = g.TRMStream(g.TRMInputParameters.from_dict(pd), nvoices=3)
    o, m = s.finish()
    assert o.shape == (3, 0)
    o, m = s.push(fr[:, :1])
    assert o.shape == (3, 0) and not np.any(m)
    dev = torch.device("cuda", 0)
    o2, n2 = s.push_device(torch.from_numpy(fr[:, 1:4]).to(dev).contiguous())
    assert n2 == o2.shape[1] > 0
    tail, _ = s.finish()
    # the same utterance again on the same stream object, host entries only: the same bits
    a = s.push(fr[:, :1])[0]; b = s.push(fr[:, 1:4])[0]; c = s.finish()[0]
    torch.cuda.synchronize()
    assert np.array_equal(b, o2.cpu().numpy()) and np.array_equal(c, tail)
    o3, n3 = s.finish_device(device=dev)
    assert n3 == 0
