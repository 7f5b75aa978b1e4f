"""Shared workload definitions: BASELINE.json configs + golden parity cases.

Values and seeds follow SURVEY.md section 8(d).  `gnuspeech.input` is the reference's own
sample control track (Applications/Monet/samples/gnuspeech.input, a DATA file: 26
utterance-rate lines + 343 frames), kept under tests/golden/ as an input fixture.
"""
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
GNUSPEECH_INPUT = os.path.join(HERE, "golden", "gnuspeech.input")


def tract_default_params():
    """TRAcT defaults (Applications/TRAcT/Controller.h:19-63, tube.c:314,327-355)."""
    return dict(outputFileFormat=0, outputRate=44100.0, controlRate=100.0, volume=60.0, channels=1,
                balance=0.0, waveform=0, tp=25.0, tnMin=5.0, tnMax=35.0, breathiness=2.5, length=17.5,
                temperature=32.0, lossFactor=0.8, apScale=5.0, mouthCoef=4000.0, noseCoef=4000.0,
                noseRadius=[0.0, 1.35, 1.7, 1.7, 1.3, 0.9], throatCutoff=1500.0, throatVol=12.0,
                usesModulation=1, mixOffset=48.0)


def monet_default_params(output_rate=44100.0):
    """Utterance-rate header of gnuspeech.input:1-26 (Monet's default male voice)."""
    return dict(outputFileFormat=0, outputRate=output_rate, controlRate=250.0, volume=60.0, channels=1,
                balance=0.0, waveform=0, tp=40.0, tnMin=16.0, tnMax=32.0, breathiness=1.0, length=17.5,
                temperature=25.0, lossFactor=0.5, apScale=3.05, mouthCoef=5000.0, noseCoef=5000.0,
                noseRadius=[0.0, 1.35, 1.96, 1.91, 1.3, 0.73], throatCutoff=1500.0, throatVol=6.0,
                usesModulation=1, mixOffset=54.0)


TRACT_VOWEL_FRAME = [-13.0, 60.0, 0.0, 0.0, 7.0, 6000.0, 1000.0,
                     0.8, 1.67, 1.905, 1.985, 0.81, 0.495, 0.73, 1.485, 0.0]


def static_frames(frame, nframes):
    return np.tile(np.asarray(frame, dtype=np.float64), (nframes, 1))


def load_gnuspeech_rows():
    """The 343 frame rows of gnuspeech.input (no doubling of the last row)."""
    rows = []
    with open(GNUSPEECH_INPUT) as f:
        lines = f.read().splitlines()
    for ln in lines[26:]:
        if ln.strip():
            rows.append([float(x) for x in ln.split()])
    return np.asarray(rows, dtype=np.float64)


def config2_frames(nvoices, nframes=251, seed=20250117):
    """Config 2: static vowels, per-voice radii U(0.4,2), velum in {0,.1,.5}, pitch U(-18,6)."""
    rng = np.random.default_rng(seed)
    fr = np.zeros((nvoices, nframes, 16), dtype=np.float64)
    radii = rng.uniform(0.4, 2.0, size=(nvoices, 8))
    velum = rng.choice([0.0, 0.1, 0.5], size=nvoices)
    pitch = rng.uniform(-18.0, 6.0, size=nvoices)
    fr[:, :, 0] = pitch[:, None]
    fr[:, :, 1] = 60.0
    fr[:, :, 4] = 7.0
    fr[:, :, 5] = 6000.0
    fr[:, :, 6] = 1000.0
    fr[:, :, 7:15] = radii[:, None, :]
    fr[:, :, 15] = velum[:, None]
    return fr


def config3_frames(nvoices, nframes=251, seed=20250118):
    """Config 3: gnuspeech.input rows tiled/cropped, cyclic offset 7k mod 343, pitch offset U(-6,6)."""
    rows = load_gnuspeech_rows()
    n = rows.shape[0]
    rng = np.random.default_rng(seed)
    poff = rng.uniform(-6.0, 6.0, size=nvoices)
    idx = (np.arange(nframes)[None, :] + (np.arange(nvoices)[:, None] * 7) % n) % n
    fr = rows[idx].copy()
    fr[:, :, 0] += poff[:, None]
    return fr


def config4_frames(nvoices, seed=20250119, lo=150, hi=1500):
    """Config 4: ragged utterances built from random windows of gnuspeech.input."""
    rows = load_gnuspeech_rows()
    n = rows.shape[0]
    rng = np.random.default_rng(seed)
    lens = rng.integers(lo, hi + 1, size=nvoices)
    out = []
    for L in lens:
        parts = []
        got = 0
        while got < L:
            w = int(rng.integers(20, 120))
            s = int(rng.integers(0, n - w))
            parts.append(rows[s:s + w])
            got += w
        out.append(np.concatenate(parts)[:L].copy())
    return out


# ---------------------------------------------------------------- golden parity cases (small)
def golden_cases():
    """name -> (params dict, frames [n,16]).  Small enough that oracle + reference run in seconds."""
    rows = load_gnuspeech_rows()
    cases = {}
    # config 1: TRAcT default static vowel, 1 s @ 44.1 kHz (101 frames x 200 samples)
    cases["tract_vowel_1s"] = (tract_default_params(), static_frames(TRACT_VOWEL_FRAME, 101))
    # Monet default voice, static vowel 0.2 s, both output rates
    mv = [-12.0, 60.0, 0.0, 0.0, 5.5, 2500.0, 500.0, 0.8, 0.89, 0.99, 0.81, 0.76, 1.05, 1.23, 0.01, 0.1]
    cases["monet_vowel_44k"] = (monet_default_params(44100.0), static_frames(mv, 51))
    cases["monet_vowel_22k"] = (monet_default_params(22050.0), static_frames(mv, 51))
    # the reference's sample utterance as the file path delivers it (last row doubled), 22.05 kHz
    full = np.concatenate([rows, rows[-1:]])
    cases["gnuspeech_input_22k"] = (monet_default_params(22050.0), full)
    # a 120-frame window with frication + aspiration at 44.1 kHz
    cases["gnuspeech_window_44k"] = (monet_default_params(44100.0), rows[60:180].copy())
    # sine waveform, modulation off
    p = monet_default_params(44100.0); p["waveform"] = 1; p["usesModulation"] = 0
    cases["sine_nomod"] = (p, rows[100:160].copy())
    # strong frication sweep: position/volume/CF/BW ramps, crossing integer tap positions
    fr = static_frames(mv, 41)
    fr[:, 1] = np.linspace(0.0, 60.0, 41)           # glottal volume through both clamps
    fr[:, 2] = np.linspace(0.0, 20.0, 41)
    fr[:, 3] = np.linspace(0.0, 45.0, 41)
    fr[:, 4] = np.linspace(0.0, 7.0, 41)
    fr[:, 5] = np.linspace(900.0, 6500.0, 41)
    fr[:, 6] = np.linspace(300.0, 3000.0, 41)
    fr[:, 0] = np.linspace(-20.0, 8.0, 41)
    fr[:, 15] = np.linspace(0.0, 1.2, 41)
    cases["frication_sweep"] = (monet_default_params(44100.0), fr)
    # short tube (child voice): tube rate 34640 Hz > 22050 Hz output -> DOWN-sampling branch
    p = monet_default_params(22050.0); p["length"] = 10.0
    cases["short_tube_downsample"] = (p, rows[100:150].copy())
    # female-ish voice 15 cm at 44.1 kHz (up-sampling, different tube rate), stereo params for writers
    p = monet_default_params(44100.0); p["length"] = 15.0; p["channels"] = 2; p["balance"] = 0.25
    p["tp"] = 35.0; p["tnMin"] = 20.0; p["tnMax"] = 40.0; p["breathiness"] = 4.0
    cases["female_15cm_stereo"] = (p, rows[200:260].copy())
    return cases


def corner_cases():
    """name -> (params dict, frames): corners no reference fixture covers, checked against the (pinned) oracle.
    pulse_no_rise_*: tp ~ 0 makes tableDiv1 = 0 -- the glottal table has no rise entries and is the fall alone
    (TRMWavetable.m:81-96); narrow_band_20hz: a frication bandwidth far below Monet's 250 Hz minimum, where
    alpha = (1/2 - beta)/2 (TRMFilters.m:16) cancels in fp32 -- pins the margin of the one-tangent band-pass form."""
    out = {}
    for tp in (0.0, 0.05):
        p = tract_default_params(); p["tp"] = tp
        out["pulse_no_rise_tp%g" % tp] = (p, static_frames(TRACT_VOWEL_FRAME, 21))
    fr = static_frames([-12.0, 54.0, 6.0, 50.0, 5.4, 2500.0, 20.0, 0.8, 0.89, 0.99, 0.81, 0.76, 0.3, 1.23, 0.9, 0.1], 41)
    out["narrow_band_20hz"] = (monet_default_params(44100.0), fr)
    return out


# ---------------------------------------------------------------- TRAcT's own loop
def tract_shim_params():
    """shim/tract_tube.c's utterance-rate globals = Applications/TRAcT/tube.c:326-352 (what the program starts with)."""
    return dict(outputFileFormat=1, outputRate=44100.0, controlRate=100.0, volume=60.0, channels=2, balance=0.0, waveform=0,
                tp=35.0, tnMin=16.0, tnMax=40.0, breathiness=2.5, length=17.0, temperature=32.0, lossFactor=0.8, apScale=2.5,
                mouthCoef=4000.0, noseCoef=4000.0, noseRadius=[1.35, 1.35, 1.7, 1.7, 1.3, 0.9], throatCutoff=1500.0,
                throatVol=6.0, usesModulation=1, mixOffset=48.0)


TRACT_SHIM_FRAME = [-0.0, 60.0, 0.0, 0.0, 8.0, 5000.0, 250.0, 0.8, 1.67, 1.905, 1.985, 0.81, 0.495, 0.73, 1.485, 0.0]
TRACT_STEP_FRAME = 69          # the control period from which radius 7 is 0.4 (as the GUI's slider would leave it)


def tract_mode_cases():
    """name -> (params, frames) run through tube.c in TRAcT's OWN loop order (oracle/ref_driver.c `tract`): the "ee" posture
    held for 68 control periods (29 988 outputs), then one radius stepped to (float)0.4 and held for 68 more."""
    fr = static_frames(TRACT_SHIM_FRAME, 138)
    fr[TRACT_STEP_FRAME:, 7 + 6] = float(np.float32(0.4))
    out = {"tract_mode_ee_step": (tract_shim_params(), fr)}
    # a held fricative posture: frication at a fractional position (two taps), aspiration, a narrow constriction --
    # the inputs on which tube.c's x10 frication taps (tube.c:1371) show
    out["tract_mode_fricative"] = (tract_shim_params(), static_frames(np.float32(TRACT_FRIC_FRAME).astype(np.float64), 70))   # (fp32-exact: what the stream receives)
    # sliders moving while it plays: the vowel, then (period TRACT_FRIC_ON) a constriction + frication + aspiration,
    # then (period TRACT_FRIC_MOVE) the frication position / centre frequency / volume move: every change STEPS
    fr = static_frames(TRACT_SHIM_FRAME, 138)
    fr[TRACT_FRIC_ON:, [2, 3, 4, 5, 6]] = np.float32([12.0, 42.5, 5.3, 3500.0, 1800.0]).astype(np.float64)
    fr[TRACT_FRIC_ON:, 7 + 5] = float(np.float32(0.2))
    fr[TRACT_FRIC_MOVE:, [3, 4, 5]] = np.float32([55.0, 6.75, 2500.0]).astype(np.float64)
    fr[TRACT_FRIC_MOVE:, 1] = 48.0
    out["tract_mode_fric_step"] = (tract_shim_params(), fr)
    return out


TRACT_SLICE = 21                # tube samples per frame of the slice cases: 1.02 ms at the shim tube's 20 600 Hz (control period 206)


def tract_slice_cases():
    """name -> (params, frames, slice): tube.c in its own loop order with the parameters changing on a grid of TRACT_SLICE
    samples -- a slider written in the MIDDLE of a control period, which whole-period pushes could only apply up to 10 ms
    late (trm_stream_set_slice; oracle/ref_driver.c `tract slice=N`).  600 slices (0.61 s): the vowel; at slice 233
    (sample 4893 = control period 23.75) radius 7 steps to 0.4; at slice 351 frication, aspiration and a constriction come
    on; at slice 437 the frication moves.  (The moves lie behind the first 8192 outputs: tests/tract_shim_driver.c places a
    slider write by filling TRAcT's circular buffer first.)"""
    fr = static_frames(TRACT_SHIM_FRAME, 601)
    fr[233:, 7 + 6] = float(np.float32(0.4))
    fr[351:, [2, 3, 4, 5, 6]] = np.float32([12.0, 42.5, 5.3, 3500.0, 1800.0]).astype(np.float64)
    fr[351:, 7 + 5] = float(np.float32(0.2))
    fr[437:, [3, 4, 5]] = np.float32([55.0, 6.75, 2500.0]).astype(np.float64)
    return {"tract_mode_slice_step": (tract_shim_params(), fr, TRACT_SLICE)}


# pitch, glotVol, aspVol, fricVol, fricPos, fricCF, fricBW, r1..r8, velum
TRACT_FRIC_FRAME = [-2.0, 54.0, 20.0, 45.0, 5.4, 4500.0, 2000.0, 0.8, 1.2, 1.5, 1.7, 1.4, 0.25, 0.9, 1.2, 0.0]
TRACT_FRIC_ON, TRACT_FRIC_MOVE = 40, 85


# ---------------------------------------------------------------- what a randomized parity run may NOT be asked to match
def bandpass_unstable(frames, tube_rate):
    """The frication band-pass (TRMFilters.m:9-29) is y = 2 (alpha (x - x2) + gamma y1 - beta y2) with
    2 beta = (1 - t) / (1 + t), t = tan(pi BW / SR): its poles leave the unit circle when t <= 0, i.e. from BW = SR / 2 on (and for BW < 0)
    (SR = the tube's sample rate).  A track that goes there makes the REFERENCE grow exponentially (outputs of 1e13 were
    seen): rounding differences are amplified without bound and no finite-precision path can match it.  Such a voice is
    outside the filter's domain and is reported as such, not compared."""
    fr = np.asarray(frames)
    return bool(fr.size) and (float(fr[:, 6].max()) >= 0.5 * float(tube_rate) or float(fr[:, 6].min()) < 0.0)


ABS_FLOOR = 1e-9     # absolute RMS below which a voice counts as matched whatever its own maximum is (a nearly silent
                     # voice: the normalisation by its maximum inflates errors of a few 1e-10; speech peaks are ~1e-3 .. 1)


def parity_error(pcm, ref_samples, ref_max):
    """(normalised RMS, absolute RMS) of a voice against the oracle; the bar is nrms <= 1e-5 OR abs <= ABS_FLOOR."""
    e = np.asarray(pcm, dtype=np.float64) - np.asarray(ref_samples, dtype=np.float64)
    a = float(np.sqrt(np.mean(e * e))) if e.size else 0.0
    return (a / ref_max if ref_max > 0 else (0.0 if a == 0.0 else np.inf)), a
