"""Control-track generation: EventList.m:883-1061 + MMDriftGenerator.m:65-78.

PARITY UNPINNED: the reference holds no event-list fixture and EventList.m needs Foundation, so the oracle
(oracle/evt_oracle.c) is checked here against hand-computed answers and an independent numpy restatement of
the drift generator; the GPU path (trm_tracks_kernel) must then equal the oracle BIT FOR BIT (the frames are
float32 values of fp64 sums built by repeated addition, which both sides perform in the same order).
"""
import ctypes as C
import itertools

import numpy as np
import pytest

import oracle_lib as O

NV = 36


def settings(micro=1, macro=1, smooth=0, drift=0, dev=1.0, cutoff=4.0, pitch=-12.0, tq=4, start=0, end=0):
    s = O.Intonation()
    s.useMicroIntonation, s.useMacroIntonation, s.useSmoothIntonation, s.useDrift = micro, macro, smooth, drift
    s.driftDeviation, s.driftCutoff, s.pitchMean, s.timeQuantization = dev, cutoff, pitch, tq
    s.startTime_ms, s.endTime_ms = start, end
    return s


def random_events(rng, n, span=40, nan_frac=0.5, smooth=False):
    """An event list shaped like Monet's: event 0 at t=0 with every tube parameter defined, later events at
    increasing multiples of 4 ms (EventList quantises insertions, EventList.m:343-372) with sparse targets."""
    if n < 2:
        return np.zeros(n, dtype=np.uint32), np.zeros((n, NV))
    times = np.concatenate([[0], np.cumsum(rng.integers(1, span // 4 + 1, size=n - 1)) * 4]).astype(np.uint32)
    vals = np.full((n, NV), np.nan)
    vals[0, :16] = rng.uniform(-2, 60, 16)
    vals[0, 32] = rng.uniform(-5, 5)
    for e in range(1, n):
        for j in range(33):
            if j < 16 or j == 32:
                if rng.random() > nan_frac:
                    vals[e, j] = rng.uniform(-2, 60) if j < 16 else rng.uniform(-8, 8)
            elif rng.random() > 0.9:
                vals[e, j] = rng.uniform(-1, 1)           # special-event offsets
        if smooth and not np.isnan(vals[e, 32]):
            vals[e, 33:36] = rng.uniform(-0.05, 0.05, 3) * [1.0, 0.1, 0.01]
    vals[n - 1, :16] = rng.uniform(-2, 60, 16)            # every parameter has a final target
    return times, vals


# ---------------------------------------------------------------- oracle against hand-computed answers (CPU)
def test_linear_ramp_known_answer():
    times = np.array([0, 20], dtype=np.uint32)
    vals = np.full((2, NV), np.nan)
    vals[0, :16] = np.arange(16.0)
    vals[1, :16] = np.arange(16.0) + 10.0
    vals[0, 32] = 0.0
    fr = O.generate_frames(times, vals, settings(macro=0, pitch=0.0))
    assert fr.shape == (5, 16)                            # t = 0, 4, 8, 12, 16; t = 20 reaches the last event
    k = np.arange(5.0)[:, None]
    want = np.arange(16.0)[None, :] + k * (10.0 / 20.0 * 4.0)
    assert np.array_equal(fr, want.astype(np.float32))


def test_nan_targets_are_skipped_and_offsets_added():
    times = np.array([0, 8, 16, 24], dtype=np.uint32)
    vals = np.full((4, NV), np.nan)
    vals[0, :16] = 1.0
    vals[3, :16] = 4.0                                    # events 1, 2 carry no target for the tube parameters
    vals[1, 16 + 7] = 0.5                                 # a special-event offset for parameter 7 appears at 8 ms ...
    vals[2, 16 + 7] = 1.5                                 # ... and heads for 1.5 at 16 ms
    vals[0, 32] = 0.0
    fr = O.generate_frames(times, vals, settings(macro=0, pitch=0.0))
    assert fr.shape == (6, 16)
    assert np.allclose(fr[:, 0], 1.0 + np.arange(6) * (3.0 / 24.0 * 4.0))
    # value 23 starts at 0 (EventList.m:928-929) and gets its first delta when event 1 is passed (t = 8):
    # (1.5 - 0) / (16 - 8) * 4 per frame until event 2 is passed at t = 16, where no later target exists -> 0
    off = np.array([0.0, 0.0, 0.0, 0.75, 1.5, 1.5])
    assert np.allclose(fr[:, 7], fr[:, 0] + off)


def test_pitch_composition_and_time_range():
    times = np.array([0, 40], dtype=np.uint32)
    vals = np.full((2, NV), np.nan)
    vals[0, :16] = 0.0
    vals[1, :16] = 0.0
    vals[0, 0], vals[1, 0] = 2.0, 2.0                     # micro-intonation value
    vals[0, 32], vals[1, 32] = 3.0, 13.0                  # contour: starts at -20 (EventList.m:959), delta from event 0's value
    full = O.generate_frames(times, vals, settings(micro=1, macro=1, pitch=-12.0))
    assert full.shape[0] == 10
    assert np.allclose(full[:, 0], 2.0 + (-20.0 + np.arange(10) * 1.0) + -12.0)
    assert np.allclose(O.generate_frames(times, vals, settings(micro=0, macro=0, pitch=-12.0))[:, 0], -12.0)
    part = O.generate_frames(times, vals, settings(start=8, end=20))
    assert np.array_equal(part, full[2:6])                # 8 <= t <= 20 (:985)


def test_smooth_intonation_cubic():
    times = np.array([0, 4, 60], dtype=np.uint32)
    vals = np.full((3, NV), np.nan)
    vals[:, :16] = 0.0
    vals[0, 32] = 1.0
    vals[1, 32:36] = [5.0, 0.5, 0.25, 0.125]              # picked up when event 1 is passed (t = 4)
    fr = O.generate_frames(times, vals, settings(micro=0, smooth=1, pitch=0.0))
    c, d33, d34, d35, want = 1.0, 0.0, 0.0, 0.0, []
    for t in range(0, 60, 4):
        want.append(c)
        d34 += d35; d33 += d34; c += d33
        if t + 4 == 4:
            c, d33, d34, d35 = 5.0, 0.5, 0.25, 0.125
    assert np.allclose(fr[:, 0], want)


def test_drift_generator_restated_in_numpy():
    f32 = np.float32
    times = np.array([0, 400], dtype=np.uint32)
    vals = np.zeros((2, NV))
    fr = O.generate_frames(times, vals, settings(micro=0, macro=0, drift=1, dev=1.5, cutoff=4.0, pitch=0.0))
    seed, prev = f32(0.7892347), f32(0.0)
    a0 = f32((np.float64(f32(4.0)) * 2.0) / np.float64(f32(250.0)))
    b1 = f32(1.0 - np.float64(a0))
    dev2, off = f32(np.float64(f32(1.5)) * 2.0), f32(1.5)
    want = []
    for _ in range(fr.shape[0]):
        temp = f32(seed * f32(377.0))
        seed = f32(temp - f32(np.int32(temp)))
        temp = f32(f32(seed * dev2) - off)
        prev = f32(f32(a0 * temp) + f32(b1 * prev))
        want.append(prev)
    assert np.array_equal(fr[:, 0], np.array(want, dtype=np.float32))


def test_frame_count_functions_agree():
    import gnuspeech_amd as g
    L = g.lib()                                            # loads without a GPU; this entry is host-side bookkeeping
    rng = np.random.default_rng(5)
    for trial in range(40):
        n = int(rng.integers(0, 12))
        times = (np.concatenate([[0], np.cumsum(rng.integers(1, 20, size=max(n - 1, 0)))]).astype(np.uint32) * (1 if trial % 3 else 4))[:n]
        for (start, end) in ((0, 0), (8, 40), (100, 10)):
            s = settings(start=start, end=end)
            a, b = C.c_size_t(), C.c_size_t()
            O.lib().trm_oracle_count_frames(times.ctypes.data, n, C.addressof(s), C.byref(a))
            s2 = g._capi.TrmIntonation.from_buffer_copy(bytes(s))
            assert L.trm_events_count_frames(times.ctypes.data, n, C.byref(s2), C.byref(b)) == 0
            assert a.value == b.value
            if n >= 2:
                vals = np.zeros((n, NV))
                assert O.generate_frames(times, vals, s).shape[0] == a.value


# ---------------------------------------------------------------- GPU == oracle, bit for bit
@pytest.fixture(scope="module")
def g():
    import gnuspeech_amd
    assert gnuspeech_amd.lib().trm_device_count() >= 1
    return gnuspeech_amd


def _to_g(gm, s):
    return gm._capi.TrmIntonation.from_buffer_copy(bytes(s))


@pytest.mark.gpu
def test_gpu_frames_equal_oracle_all_switches(g):
    import cases
    b = g.TRMBatch(g.TRMInputParameters.from_dict(cases.monet_default_params()))
    rng = np.random.default_rng(11)
    for micro, macro, smooth, drift in itertools.product((0, 1), repeat=4):
        times, vals = random_events(rng, 25, smooth=bool(smooth))
        s = settings(micro, macro, smooth, drift, dev=0.8, cutoff=3.0, pitch=-9.5)
        want = O.generate_frames(times, vals, s)
        el = g.EventList(pitch_mean=-9.5)
        for t, v in zip(times, vals):
            e = g.Event(t)
            e.values[:] = v
            el.events.append(e)
        el.intonation.shouldUseMicroIntonation, el.intonation.shouldUseMacroIntonation = micro, macro
        el.intonation.shouldUseSmoothIntonation, el.intonation.shouldUseDrift = smooth, drift
        el.intonation.driftDeviation, el.intonation.driftCutoff = 0.8, 3.0
        got = el.generateOutputInTimeRange(b)
        assert got.shape == want.shape
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (micro, macro, smooth, drift)


@pytest.mark.gpu
def test_gpu_frames_equal_oracle_irregular_event_times(g):
    """Event times that are not multiples of the 4 ms frame interval, several events inside one interval (the
    generator advances one event per frame, so it runs late and the unsigned `time - currentTime` of
    EventList.m:1040-1041 wraps), and time ranges that cut the utterance."""
    import cases
    b = g.TRMBatch(g.TRMInputParameters.from_dict(cases.monet_default_params()))
    rng = np.random.default_rng(41)
    for trial in range(12):
        n = int(rng.integers(2, 30))
        times = np.concatenate([[0], np.cumsum(rng.integers(0, 11, size=n - 1))]).astype(np.uint32)     # incl. equal times
        _, vals = random_events(rng, n, smooth=bool(trial & 1))
        start, end = (0, 0) if trial % 3 else (int(rng.integers(0, 40)), int(rng.integers(40, 200)))
        s = settings(1, 1, trial & 1, (trial >> 1) & 1, dev=1.2, cutoff=6.0, pitch=-7.25, start=start, end=end)
        want = O.generate_frames(times, vals, s)
        st = b.prepare_events_device([(times, vals)], _to_g(g, s))
        b.generate_frames_device(st)
        import torch
        torch.cuda.synchronize()
        got = st["frames"].cpu().numpy()[:want.shape[0]]
        assert int(st["nframes_generated"].cpu().numpy()[0]) == want.shape[0]
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), trial


@pytest.mark.gpu
def test_gpu_ragged_batch_events_to_pcm(g):
    """Event lists -> frames -> PCM without leaving the device: the generated frames equal the oracle's bit for
    bit, and the tube driven by them produces the same bits as the tube driven by uploaded frames."""
    import torch
    import cases
    pd = cases.monet_default_params(22050.0)
    b = g.TRMBatch(g.TRMInputParameters.from_dict(pd))
    rng = np.random.default_rng(23)
    lists = [random_events(rng, int(n), span=24) for _ in range(9) for n in (2, 3, 17, 40, 9, 1, 0, 30)]   # 72 voices, incl. < 2 events
    for i, (t, v) in enumerate(lists):                                                          # speech-like ranges
        v[:, 0] = np.where(np.isnan(v[:, 0]), np.nan, np.clip(v[:, 0], -2, 2))
        v[:, 1:4] = np.where(np.isnan(v[:, 1:4]), np.nan, np.clip(v[:, 1:4], 0, 60))
        v[:, 4] = np.where(np.isnan(v[:, 4]), np.nan, np.clip(v[:, 4] / 10, 0, 7))
        v[:, 5:7] = np.where(np.isnan(v[:, 5:7]), np.nan, 500 + 50 * v[:, 5:7])
        v[:, 7:16] = np.where(np.isnan(v[:, 7:16]), np.nan, 0.1 + np.abs(v[:, 7:16]) / 30)
        v[:, 16:32] = np.nan
    s = settings(1, 1, 0, 1, dev=0.5, cutoff=2.0, pitch=-12.0, start=8, end=2000)
    st = b.prepare_events_device(lists, _to_g(g, s))
    b.generate_frames_device(st)
    torch.cuda.synchronize()
    frames = st["frames"].cpu().numpy()
    ngen = st["nframes_generated"].cpu().numpy()
    foff = st["frame_offset"].cpu().numpy()
    per_voice = []
    for v, (t, val) in enumerate(lists):
        want = O.generate_frames(t, val, s) if len(t) >= 2 else np.zeros((0, 16), np.float32)
        assert ngen[v] == want.shape[0] == st["nframes_host"][v]
        got = frames[foff[v]:foff[v] + ngen[v]]
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), "voice %d" % v
        per_voice.append(want)
    b.synthesize_device(st)
    torch.cuda.synchronize()
    pcm_a = st["out"].cpu().numpy().copy()
    st2 = b.prepare_device(per_voice)
    b.synthesize_device(st2)
    torch.cuda.synchronize()
    assert np.array_equal(st["number_samples"].cpu().numpy(), st2["number_samples"].cpu().numpy())
    assert st["out_alloc"] == st2["out_alloc"] and np.all(np.isfinite(pcm_a[:st["out_alloc"]]))
    assert np.array_equal(pcm_a[:st["out_alloc"]], st2["out"].cpu().numpy()[:st2["out_alloc"]])


def test_drift_seed_carries_over_between_utterances():
    """The reference keeps ONE drift generator per EventList; only -init sets its seed, -configure... resets the filter
    memory but "seed is not changed" (MMDriftGenerator.m:27-58, EventList.m:105-106, 903): the second utterance of a
    list continues the noise sequence.  trm_intonation.driftSeed + trm_drift_seed_after() give a caller that: the
    second utterance generated with the carried seed equals the numpy restatement continued across both."""
    import gnuspeech_amd as g
    f32 = np.float32
    L = g.lib()                                            # host-side bookkeeping: loads without a GPU
    times = np.array([0, 400], dtype=np.uint32)
    vals = np.zeros((2, NV))
    s1 = settings(micro=0, macro=0, drift=1, dev=1.5, cutoff=4.0, pitch=0.0)
    a = O.generate_frames(times, vals, s1)
    seed_after = L.trm_drift_seed_after(0.0, len(a))
    s2 = settings(micro=0, macro=0, drift=1, dev=1.5, cutoff=4.0, pitch=0.0)
    s2.driftSeed = seed_after
    b = O.generate_frames(times, vals, s2)
    # numpy: one generator, two utterances (the filter memory is cleared by -configure..., the seed is not)
    seed = f32(0.7892347)
    a0 = f32((np.float64(f32(4.0)) * 2.0) / np.float64(f32(250.0)))
    b1 = f32(1.0 - np.float64(a0))
    dev2, off = f32(np.float64(f32(1.5)) * 2.0), f32(1.5)
    both = []
    for utt in range(2):
        prev, want = f32(0.0), []
        for _ in range(len(a)):
            temp = f32(seed * f32(377.0))
            seed = f32(temp - f32(np.int32(temp)))
            temp = f32(f32(seed * dev2) - off)
            prev = f32(f32(a0 * temp) + f32(b1 * prev))
            want.append(prev)
        both.append(np.array(want, dtype=np.float32))
        if utt == 0:
            assert f32(seed_after) == seed
    assert np.array_equal(a[:, 0], both[0]) and np.array_equal(b[:, 0], both[1])
    assert not np.array_equal(a[:, 0], b[:, 0])


@pytest.mark.gpu
def test_gpu_drift_seed_carries_over(g):
    """The device generator with a carried seed == the oracle with the same seed, bit for bit; gnuspeech_amd.EventList
    carries it by itself like the reference's EventList (second utterance of the same list)."""
    import cases
    b = g.TRMBatch(g.TRMInputParameters.from_dict(cases.monet_default_params()))
    rng = np.random.default_rng(23)
    times, vals = random_events(rng, 25, smooth=False)
    el = g.EventList(pitch_mean=-9.5)
    for t, v in zip(times, vals):
        e = g.Event(t)
        e.values[:] = v
        el.events.append(e)
    el.intonation.shouldUseSmoothIntonation = False
    el.intonation.driftDeviation, el.intonation.driftCutoff = 0.8, 3.0
    first = el.generateOutputInTimeRange(b)
    assert el.driftSeed != 0.0
    second = el.generateOutputInTimeRange(b)
    s = settings(1, 1, 0, 1, dev=0.8, cutoff=3.0, pitch=-9.5)
    want1 = O.generate_frames(times, vals, s)
    s.driftSeed = g.lib().trm_drift_seed_after(0.0, len(want1))
    want2 = O.generate_frames(times, vals, s)
    assert np.array_equal(first.view(np.uint32), want1.view(np.uint32))
    assert np.array_equal(second.view(np.uint32), want2.view(np.uint32))
    assert not np.array_equal(first, second)
