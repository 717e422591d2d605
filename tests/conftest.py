import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
TESTS = os.path.join(ROOT, "tests")
if TESTS not in sys.path:  # helper modules next to the tests (util, gradcam_picture)
    sys.path.insert(1, TESTS)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


@pytest.fixture
def lib_option():
    """lib_option(name, value): one of the library's measurement switches (cnn_amd_set_option; the CNN_AMD_* environment is only
    read once, when the library is first used) for the duration of a test; value None removes it.  Restored afterwards."""
    from cnn_amd import capi

    saved = {}

    def set_(name, value):
        if name not in saved:
            saved[name] = capi.get_option(name)
        capi.set_option(name, value)

    yield set_
    for name, old in saved.items():
        capi.set_option(name, old)
