import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

REFERENCE = "/root/reference"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: longer CPU test")


@pytest.fixture(scope="session")
def reference_util_h():
    """Text of the reference's util.h, for value checks of the generated tables.
    Only present in the build container; tests using it skip elsewhere."""
    p = os.path.join(REFERENCE, "corintho_ai/cpp/include/util.h")
    if not os.path.exists(p):
        pytest.skip("reference tree not mounted")
    with open(p) as f:
        return f.read()
