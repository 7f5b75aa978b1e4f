#!/usr/bin/env python3
"""Regenerates tests/golden/*.npz from the REFERENCE's own C tube.

Runs oracle/_ref/tube_ref (Applications/TRAcT/tube.c compiled in place from /root/reference
by oracle/Makefile, driven by oracle/ref_driver.c) on every case of tests/cases.golden_cases()
and freezes what the reference computed: tube-rate doubles, converter output (fp32, as tube.c's
dataEmpty emits it), numberSamples, maximumSampleValue, FIR taps, derived constants.
cases.tract_mode_cases() run through the same binary in TRAcT's own loop order (ref_driver.c `tract`),
cases.tract_slice_cases() in that order with the parameters changing on a grid of samples (`tract slice=N`).
A list of case names on the command line regenerates only those.
Only runs where /root/reference exists; the .npz files are the committed fixtures.

    make -C oracle && python tests/golden/make_golden.py
"""
import json
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import cases  # noqa: E402
import oracle_lib as O  # noqa: E402


def main():
    if not O.have_ref():
        sys.exit("oracle/_ref/tube_ref missing: run `make -C oracle` where /root/reference exists")
    h_saved = False
    todo = [(n, c + (0,), False) for n, c in cases.golden_cases().items()] + [(n, c + (0,), True) for n, c in cases.tract_mode_cases().items()]
    todo += [(n, c, True) for n, c in cases.tract_slice_cases().items()]       # (round 4: parameters on a grid of `slice` samples)
    only = sys.argv[1:]
    for name, (pd, frames, slc), tract in todo:
        if only and name not in only:
            continue
        p = O.InputParams.from_dict(pd)
        with tempfile.TemporaryDirectory() as d:
            r = O.run_ref(p, frames, d, tract=tract, slice=slc)
        np.savez_compressed(
            os.path.join(HERE, name + ".npz"),
            params_json=np.array(json.dumps(pd)),
            frames=np.asarray(frames, dtype=np.float64),
            tubeSamples=r["tubeSamples"],
            samples_f32=r["samples_f32"],
            numberSamples=np.int64(r["numberSamples"]),
            maximumSampleValue=np.float64(r["maximumSampleValue"]),
            firCoef=r["firCoef"],
            derived=np.array([r["controlPeriod"], r["sampleRate"], r["padSize"], r["firTaps"],
                              r["timeRegisterIncrement"], r["phaseIncrement"]], dtype=np.int64),
            tap_err=np.float64(r["tap_err"]),
            slice=np.int64(slc),
        )
        if not h_saved:
            np.savez_compressed(os.path.join(HERE, "src_tables.npz"), h=r["h"], deltaH=r["deltaH"])
            h_saved = True
        print("%-26s frames=%4d tube=%6d out=%6d max=%.6g" % (
            name, len(frames), len(r["tubeSamples"]), r["numberSamples"], r["maximumSampleValue"]))


if __name__ == "__main__":
    main()
