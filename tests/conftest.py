import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")
    # MRS_FUZZ_SEED_OFFSET=k: every numpy generator seeded with an integer gets seed + k, i.e. the differential tests run on
    # other random inputs (a few tests pin facts of their fixed inputs and are expected to object; the rest must still pass)
    off = int(os.environ.get("MRS_FUZZ_SEED_OFFSET", "0"))
    if off:
        import numpy as np
        orig = np.random.default_rng
        np.random.default_rng = lambda seed=None, *a, **k: orig(seed + off if isinstance(seed, int) else seed, *a, **k)


@pytest.fixture(scope="session")
def oracle():
    """The CPU checker (oracle/), built on demand.  Test infrastructure only."""
    from oracle import pyoracle
    pyoracle.lib()
    return pyoracle


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
