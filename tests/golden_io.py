"""Loader for the committed reference-generated fixtures in tests/golden/*.npz."""
import json
import os

import numpy as np

import oracle_lib as O

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASE_NAMES = ["tract_vowel_1s", "monet_vowel_44k", "monet_vowel_22k", "gnuspeech_input_22k",
              "gnuspeech_window_44k", "sine_nomod", "frication_sweep", "short_tube_downsample",
              "female_15cm_stereo"]
# the reference's tube.c stepped in TRAcT's OWN loop order (oracle/ref_driver.c `tract`)
TRACT_CASE_NAMES = ["tract_mode_ee_step", "tract_mode_fricative", "tract_mode_fric_step"]
# ... and with its parameters changing on a grid of `slice` samples (`tract slice=N`; trm_stream_set_slice)
TRACT_SLICE_CASE_NAMES = ["tract_mode_slice_step"]


def load(name):
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"), allow_pickle=False)
    pd = json.loads(str(z["params_json"]))
    g = {k: z[k] for k in z.files if k != "params_json"}
    g["params_dict"] = pd
    g["params"] = O.InputParams.from_dict(pd)
    g["numberSamples"] = int(g["numberSamples"])
    g["maximumSampleValue"] = float(g["maximumSampleValue"])
    g["slice"] = int(g["slice"]) if "slice" in g else 0
    return g
