"""Oracle against the reference's C tube RUN LIVE (only where oracle/_ref/tube_ref was built,
i.e. where /root/reference exists).  Extends the frozen fixtures with fresh random cases."""
import numpy as np
import pytest

import cases
import oracle_lib as O

pytestmark = pytest.mark.skipif(not O.have_ref(), reason="oracle/_ref/tube_ref not built (no /root/reference here)")


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_random_tracks_bit_exact(oracle, tmp_path, seed):
    rng = np.random.default_rng(seed)
    pd = cases.monet_default_params(44100.0 if seed % 2 else 22050.0)
    pd["length"] = float(rng.uniform(11.0, 19.0))
    pd["temperature"] = float(rng.uniform(25.0, 38.0))
    pd["breathiness"] = float(rng.uniform(0.0, 8.0))
    pd["lossFactor"] = float(rng.uniform(0.1, 3.0))
    rows = cases.load_gnuspeech_rows()
    s = int(rng.integers(0, 280))
    fr = rows[s:s + 40].copy()
    fr[:, 0] += rng.uniform(-6, 6)
    fr[:, 3] += rng.uniform(0, 30, size=40)          # audible frication
    fr[:, 4] = rng.uniform(0, 7, size=40)
    p = oracle.InputParams.from_dict(pd)
    o = oracle.synthesize(p, fr, keep_tube=True)
    r = oracle.run_ref(p, fr, str(tmp_path))
    assert o["numberSamples"] == r["numberSamples"]
    assert o["maximumSampleValue"] == r["maximumSampleValue"]
    assert np.array_equal(o["tubeSamples"], r["tubeSamples"])
    assert np.array_equal(o["samples"].astype(np.float32), r["samples_f32"])
