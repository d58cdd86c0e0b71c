import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


@pytest.fixture(scope="session")
def ora():
    """The CPU oracle (test infrastructure)."""
    from oracle import oracle
    return oracle.api()


@pytest.fixture(scope="session")
def gpu():
    """The product: librdf_mi355x.so through its C ABI.  Fails loudly if the library is missing."""
    from rust_dataframe_amd import lib
    api = lib.api()  # raises ImportError when the .so is absent: no fallback
    if lib.device_count() < 1:
        pytest.fail("GPU test selected but no HIP device is visible")
    return api
