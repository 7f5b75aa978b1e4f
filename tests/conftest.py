import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # A fresh checkout has no built library (artefacts are not in history): build it once, as __graft_entry__.build()
    # does (hipcc cross-compiles gfx950 without a GPU).  An existing library is used as it is.
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if not os.path.exists(os.path.join(root, "gnuspeech_amd", "libtrm_hip.so")):
        import subprocess
        subprocess.check_call(["make", "-s", "-C", os.path.join(root, "gnuspeech_amd", "csrc")])


@pytest.fixture(scope="session")
def oracle():
    import oracle_lib
    oracle_lib.lib()
    return oracle_lib
