import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


# The run-time compiler keeps its code objects across processes (~/.cache/rdf_mi355x/jit by default).  A test session gets a
# directory of its own, so that "met for the first time" means the same thing on a fresh box and on one that ran the suite before.
import atexit  # noqa: E402
import shutil  # noqa: E402
import tempfile  # noqa: E402
if "RDF_JIT_CACHE" not in os.environ:          # (only then: a directory is created, and removed again when the session ends)
    _jit_dir = tempfile.mkdtemp(prefix="rdf_jit_cache_")
    os.environ["RDF_JIT_CACHE"] = _jit_dir
    atexit.register(shutil.rmtree, _jit_dir, ignore_errors=True)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


@pytest.fixture(scope="session")
def ora():
    """The CPU oracle (test infrastructure)."""
    from oracle import oracle
    return oracle.api()


@pytest.fixture(params=["spec", "interp"])
def gpu(request):
    """The product: librdf_mi355x.so through its C ABI.  Fails loudly if the library is missing.
    Every GPU test runs twice: with the ahead-of-time specialised kernels enabled (default) and with the
    general evaluator forced, so both device paths are held to the same parity bar."""
    from rust_dataframe_amd import lib
    api = lib.api()  # raises ImportError when the .so is absent: no fallback
    if lib.device_count() < 1:
        pytest.fail("GPU test selected but no HIP device is visible")
    lib.set_option("spec", 1 if request.param == "spec" else 0)
    lib.set_option("fast_filter", 1 if request.param == "spec" else 0)
    # shapes outside the catalogs are compiled at run time by default (rdf_jit.cpp, about a second each): the parity suites walk
    # hundreds of such shapes and hold them to the interpreter, test_kernels_compiled_at_run_time turns the compiler on
    lib.set_option("jit", 2 if os.environ.get("RDF_TEST_JIT") == "1" and request.param == "spec" else 0)   # RDF_TEST_JIT=1: the whole suite through the run-time compiler (2 = every call waits for its kernel; minutes of compiles)
    yield api
    lib.set_option("spec", 1)
    lib.set_option("fast_filter", 1)
    lib.set_option("jit", 1)
