"""Host model of the kernels' arithmetic (tests/_emul/trm_emul.cc: gnuspeech_amd/csrc/trm_lane.h and trm_quad.h
compiled for the CPU, TEST INFRASTRUCTURE).  Without a GPU this checks
  * that the four-lane tube step (two-wide junction rounds, cross-part moves, both ends in one part) reproduces the
    one-voice-per-lane tube_step BIT FOR BIT on random states and coefficients (the data movement is exact), and
  * that both formulations (fp32 signal path, closed-form tracks, prefix-sum oscillator phase, direct-form FIR)
    meet the parity tolerance against the oracle on the reference fixtures."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import golden_io
import oracle_lib as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "_emul", "trm_emul.cc")
LIB = os.path.join(ROOT, "tests", "_emul", "libtrm_emul.so")


@pytest.fixture(scope="module")
def emul():
    csrc = os.path.join(ROOT, "gnuspeech_amd", "csrc")
    deps = [SRC] + [os.path.join(csrc, f) for f in ("trm_lane.h", "trm_quad.h", "trm_oct.h", "trm_setup.cc", "trm_setup.h")]
    if not os.path.exists(LIB) or any(os.path.getmtime(d) > os.path.getmtime(LIB) for d in deps):
        subprocess.check_call(["g++", "-O2", "-ffp-contract=off", "-fPIC", "-shared", "-std=c++17", "-o", LIB, SRC,
                               os.path.join(csrc, "trm_setup.cc"), "-lm"])
    E = C.CDLL(LIB)
    sig = [C.POINTER(O.InputParams), C.POINTER(C.c_float), C.c_size_t, C.POINTER(C.c_float), C.c_size_t,
           C.POINTER(C.c_uint32), C.POINTER(C.c_float), C.POINTER(C.c_float)]
    E.trm_emul_synthesize.argtypes = sig
    E.trm_emul_synthesize_quad.argtypes = sig
    E.trm_emul_synthesize_tract.argtypes = sig
    E.trm_emul_quad_selfcheck.argtypes = [C.POINTER(O.InputParams), C.c_int, C.c_uint]
    E.trm_emul_oct_selfcheck.argtypes = [C.POINTER(O.InputParams), C.c_int, C.c_uint]
    return E


def test_four_lane_step_equals_one_lane_step_bit_for_bit(emul):
    p = golden_io.load("gnuspeech_window_44k")["params"]
    for seed in (1, 2, 3):
        assert emul.trm_emul_quad_selfcheck(C.byref(p), 20000, seed) == 0


def test_eight_lane_step_equals_one_lane_step_bit_for_bit(emul):
    """trm_oct.h: eight parts of two junction slots per voice, neighbours by row rotation.  The host model poisons what a
    rotation brings in from outside the voice's eight lanes (NaN), so a use of it fails here."""
    p = golden_io.load("gnuspeech_window_44k")["params"]
    for seed in (1, 2, 3):
        assert emul.trm_emul_oct_selfcheck(C.byref(p), 20000, seed) == 0


@pytest.mark.parametrize("name", ["tract_vowel_1s", "gnuspeech_input_22k", "sine_nomod", "frication_sweep", "female_15cm_stereo"])
def test_both_formulations_against_oracle(emul, name):
    g = golden_io.load(name)
    p, fr = g["params"], np.ascontiguousarray(g["frames"], dtype=np.float32)
    o = O.synthesize(p, fr.astype(np.float64))
    for fn in (emul.trm_emul_synthesize, emul.trm_emul_synthesize_quad):
        cap = len(fr) * 700 + 2000
        out = np.zeros(cap, dtype=np.float32)
        n, m = C.c_uint32(), C.c_float()
        assert fn(C.byref(p), fr.ctypes.data_as(C.POINTER(C.c_float)), len(fr), out.ctypes.data_as(C.POINTER(C.c_float)), cap,
                  C.byref(n), C.byref(m), None) == 0
        assert n.value == o["numberSamples"]
        e = (out[:n.value].astype(np.float64) - o["samples"]) / o["maximumSampleValue"]
        assert float(np.sqrt(np.mean(e * e))) <= 1e-5
        assert abs(m.value - o["maximumSampleValue"]) / o["maximumSampleValue"] < 2e-4


@pytest.mark.parametrize("name", ["pulse_no_rise_tp0", "pulse_no_rise_tp0.05", "narrow_band_20hz"])
def test_corner_cases_against_oracle(emul, name):
    """tests/cases.py corner_cases: a glottal pulse without a rise (tp ~ 0: tableDiv1 == 0, ADVICE r03 -- the select-free
    table returned silence there) and a 20 Hz frication band, device arithmetic on the host against the oracle."""
    import cases
    pd, frames = cases.corner_cases()[name]
    p = O.InputParams.from_dict(pd)
    fr = np.ascontiguousarray(frames, dtype=np.float32)
    o = O.synthesize(p, fr.astype(np.float64))
    assert o["maximumSampleValue"] > 1e-4
    for fn in (emul.trm_emul_synthesize, emul.trm_emul_synthesize_quad):
        cap = len(fr) * 700 + 2000
        out = np.zeros(cap, dtype=np.float32)
        n, m = C.c_uint32(), C.c_float()
        assert fn(C.byref(p), fr.ctypes.data_as(C.POINTER(C.c_float)), len(fr), out.ctypes.data_as(C.POINTER(C.c_float)), cap,
                  C.byref(n), C.byref(m), None) == 0
        assert n.value == o["numberSamples"]
        e = (out[:n.value].astype(np.float64) - o["samples"]) / o["maximumSampleValue"]
        assert float(np.sqrt(np.mean(e * e))) <= 1e-5, float(np.sqrt(np.mean(e * e)))


@pytest.mark.parametrize("name", golden_io.TRACT_CASE_NAMES)
def test_tract_order_arithmetic_against_the_reference(emul, name):
    """The device arithmetic in TRAcT's loop order (TRM_STREAM_MODE_TRACT: held parameters that step, x10 frication
    amplitude, x100) on the host, against what the REFERENCE's tube.c produced in that order (tests/golden/tract_mode_*):
    the whole utterance, stepped periods and fricative stretches included, at the one tolerance."""
    g = golden_io.load(name)
    p, fr = g["params"], np.ascontiguousarray(g["frames"], dtype=np.float32)
    cap = len(fr) * 700 + 2000
    out = np.zeros(cap, dtype=np.float32)
    n, m = C.c_uint32(), C.c_float()
    assert emul.trm_emul_synthesize_tract(C.byref(p), fr.ctypes.data_as(C.POINTER(C.c_float)), len(fr),
                                          out.ctypes.data_as(C.POINTER(C.c_float)), cap, C.byref(n), C.byref(m), None) == 0
    assert n.value == g["numberSamples"]
    e = (out[:n.value].astype(np.float64) - g["samples_f32"].astype(np.float64)) / g["maximumSampleValue"]
    assert float(np.sqrt(np.mean(e * e))) <= 1e-5


def test_product_fir_table_equals_reference_taps(emul):
    """The product carries the oscillator FIR as a constant table (TRMFIRFilter.h:7-9 fixes the design's inputs); it must
    be what the REFERENCE binary computed (firCoef of every fixture), to the last bit of the double."""
    half = np.zeros(25)
    emul.trm_emul_fir_half.argtypes = [C.POINTER(C.c_double)]
    emul.trm_emul_fir_half(half.ctypes.data_as(C.POINTER(C.c_double)))
    for name in golden_io.CASE_NAMES:
        ref = golden_io.load(name)["firCoef"]
        assert len(ref) == 49
        assert np.array_equal(half, ref[:25]) and np.array_equal(half[:24], ref[:24:-1])
