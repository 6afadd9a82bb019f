import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _native_built():
    """Build the native pieces if they are absent (hipcc cross-compiles without a GPU).
    On the GPU box the prebuilt .so files travel with the snapshot and are used as they are."""
    import __graft_entry__
    _build = __graft_entry__._load_builder()
    if not (os.path.exists(_build.HIP_LIB) and os.path.exists(_build.EXT_LIB)):
        _build.build_all()
    from oracle import oracle
    oracle.build()


@pytest.fixture(scope="session")
def golden():
    import json
    with open(os.path.join(ROOT, "tests", "golden", "reference_test_rw.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def libc_vectors():
    import json
    with open(os.path.join(ROOT, "tests", "golden", "libc_rand.json")) as f:
        return json.load(f)
