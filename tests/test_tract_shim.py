"""shim/tract_tube.c: TRAcT's tube.h interface over a one-voice stream in TRM_STREAM_MODE_TRACT.  The C
shim is compiled with gcc and driven like Controller.m drives tube.c (initializeSynthesizer, getCircBuff2, parameter
writes through the getter pointers); what comes out of its circular buffer is compared with the REFERENCE's tube.c run
in its own loop order (tests/golden/tract_mode_*.npz, oracle/ref_driver.c `tract`) over the WHOLE utterance -- slider
moves, fricative stretches, x10 frication taps, x100 gain included -- at the one tolerance, 1e-5."""
import os
import subprocess

import numpy as np
import pytest

import cases
import golden_io

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RING = 8192                # shim/tract_tube.c CIRC_BUFF2_SIZE == tube.c's


@pytest.fixture(scope="module")
def driver(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("shim") / "tract_shim_driver")
    subprocess.check_call(["gcc", "-O2", "-o", exe, os.path.join(ROOT, "tests", "tract_shim_driver.c"),
                           os.path.join(ROOT, "shim", "tract_tube.c"), "-L" + os.path.join(ROOT, "gnuspeech_amd"),
                           "-l:libtrm_hip.so", "-Wl,-rpath," + os.path.join(ROOT, "gnuspeech_amd"), "-lpthread", "-lm"])
    return exe


def run_driver(exe, tmp_path, total, *events, slice_samples=0):
    """slice_samples: TRACT_SLICE_SAMPLES for the shim -- 0 = one control period per push (moves land at period boundaries:
    what the period-stepped goldens need), None = the shim's default (a millisecond)."""
    out = str(tmp_path / "heard.f32")
    env = dict(os.environ, LD_LIBRARY_PATH="/opt/rocm/lib:" + os.environ.get("LD_LIBRARY_PATH", ""))
    env.pop("TRACT_SLICE_SAMPLES", None)
    if slice_samples is not None:
        env["TRACT_SLICE_SAMPLES"] = str(slice_samples)
    r = subprocess.run([exe, out, str(total)] + list(events), capture_output=True, text=True, env=env, timeout=120)
    assert r.returncode == 0, r.stderr
    return np.fromfile(out, dtype=np.float32), r


def nrms(x, ref, mx):
    e = (np.asarray(x, dtype=np.float64) - np.asarray(ref, dtype=np.float64)) / mx
    return float(np.sqrt(np.mean(e * e)))


def outputs_after(periods, gold):
    """Converter outputs the stream has returned after `periods` pushes (control periods, or the golden's slices:
    trm_stream_samples_for_push's sum)."""
    cp, inc = gold["slice"] or int(gold["derived"][0]), int(gold["derived"][4])
    n = periods * cp
    return 0 if n == 0 else ((n << 16) - 1) // inc + 1


def move_at(period, gold, **values):
    """The driver event that makes `values` the parameters of control period `period` (0-based) and later: written while
    the synthesis thread is blocked in the middle of period - 1, RING samples ahead of what has been heard."""
    mid = (outputs_after(period - 1, gold) + outputs_after(period, gold)) // 2
    assert mid - RING > 0
    return "%d:%s" % (mid - RING, ",".join("%s=%r" % (k, float(v)) for k, v in values.items()))


FRAME_NAMES = ["glotPitch", "glotVol", "aspVol", "fricVol", "fricPos", "fricCF", "fricBW"] + ["r%d" % i for i in range(8)] + ["velum"]


def events_from_frames(gold):
    """Slider moves that turn the shim's start-up posture into the golden's frame sequence: frame f is what control
    period f - 1 runs on (oracle/ref_driver.c `tract`)."""
    fr = gold["frames"].astype(np.float32).astype(np.float64)         # (the shim hands the library floats)
    ev = []
    start = np.float32(cases.TRACT_SHIM_FRAME).astype(np.float64)
    if not np.array_equal(fr[1], start):
        ev.append("-1:" + ",".join("%s=%r" % (FRAME_NAMES[i], float(fr[1][i])) for i in range(16) if fr[1][i] != start[i]))
    for f in range(2, len(fr)):
        ch = [i for i in range(16) if fr[f][i] != fr[f - 1][i]]
        if ch:
            ev.append(move_at(f - 1, gold, **{FRAME_NAMES[i]: fr[f][i] for i in ch}))
    return ev


@pytest.mark.parametrize("name", golden_io.TRACT_CASE_NAMES)
def test_tract_shim_against_the_reference_in_tract_order(driver, tmp_path, name):
    gold = golden_io.load(name)
    assert gold["params_dict"] == cases.tract_shim_params()            # the shim's utterance-rate globals
    want, mx = gold["samples_f32"].astype(np.float64), gold["maximumSampleValue"]
    # everything the golden holds before its converter's flush (the shim never ends its utterance)
    total = outputs_after(len(gold["frames"]) - 1, gold) - 64
    ev = events_from_frames(gold)
    assert len(ev) == {"tract_mode_ee_step": 1, "tract_mode_fricative": 1, "tract_mode_fric_step": 2}[name]
    heard, r = run_driver(driver, tmp_path, total, *ev)
    assert heard.size == total and np.all(np.isfinite(heard))
    assert "controlPeriod %d" % int(gold["derived"][0]) in r.stdout
    e = nrms(heard, want[:total], mx)
    assert e <= 1e-5, "%s: normalised RMS %.3e over the whole utterance" % (name, e)
    # and no stretch of it hides behind the average: every control period on its own (the stepped ones included)
    per = int(round(44100.0 / 100.0))
    worst = max(nrms(heard[i:i + per], want[i:i + per], mx) for i in range(0, total - per, per))
    assert worst <= 1e-5, "%s: worst control period %.3e" % (name, worst)


def test_tract_shim_slider_write_lands_within_a_millisecond(driver, tmp_path):
    """The shim as it ships pushes a frame per millisecond (21 tube samples at its 20 600 Hz): slider writes in the MIDDLE of
    a control period -- radius 7 at sample 4893 (period 23.75), a fricative at slice 351, a move at 437 -- are heard from the
    next slice on, like tube.c hears them from the next sample on.  Against the REFERENCE's tube.c run with its parameters
    changing on that grid (tests/golden/tract_mode_slice_step.npz, oracle/ref_driver.c `tract slice=21`): the whole
    utterance and every single millisecond of it."""
    gold = golden_io.load("tract_mode_slice_step")
    assert gold["slice"] == cases.TRACT_SLICE and gold["params_dict"] == cases.tract_shim_params()
    want, mx = gold["samples_f32"].astype(np.float64), gold["maximumSampleValue"]
    total = outputs_after(len(gold["frames"]) - 1, gold) - 64
    ev = events_from_frames(gold)
    assert len(ev) == 3
    heard, r = run_driver(driver, tmp_path, total, *ev, slice_samples=None)
    assert "slice %d" % cases.TRACT_SLICE in r.stdout
    assert heard.size == total and np.all(np.isfinite(heard))
    assert nrms(heard, want[:total], mx) <= 1e-5
    per = 45                                # a slice's outputs
    worst = max(nrms(heard[i:i + per], want[i:i + per], mx) for i in range(0, total - per, per))
    assert worst <= 1e-5, "worst millisecond %.3e" % worst
    # with whole control periods per push the same writes come up to 10 ms late: far outside the tolerance
    late, _ = run_driver(driver, tmp_path, total, *ev, slice_samples=0)
    assert nrms(late, want[:total], mx) > 1e-3


def test_tract_shim_equals_the_python_stream(driver, tmp_path):
    """The C shim and gnuspeech_amd.TRMStream(mode="tract") are the same calls: bit for bit."""
    import gnuspeech_amd as g
    gold = golden_io.load("tract_mode_fric_step")
    total = 40000
    heard, _ = run_driver(driver, tmp_path, total, *events_from_frames(gold))
    s = g.TRMStream(g.TRMInputParameters.from_dict(cases.tract_shim_params()), nvoices=1, mode="tract")
    fr = gold["frames"].astype(np.float32)
    parts, n = [], 0
    for f in range(1, len(fr)):
        o, _ = s.push(fr[f:f + 1])
        parts.append(o[0])
        n += o.shape[1]
        if n >= total:
            break
    assert np.array_equal(heard, np.concatenate(parts)[:total])
