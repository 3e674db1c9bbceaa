import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _bounded_checker_threads():
    """The CPU checkers (oracle, oracle/_ref) are OpenMP code; on a 256-thread host their loops over the small
    frames of these tests spend seconds per call forking and joining.  32 threads keep the full-frame checks fast
    and the small ones cheap.  (bench.py's cpu_baseline is a separate process and uses every core.)"""
    import ctypes
    if (os.cpu_count() or 1) > 32 and "OMP_NUM_THREADS" not in os.environ:
        try:
            ctypes.CDLL("libgomp.so.1").omp_set_num_threads(32)  # the oracle (gcc)
        except OSError:
            pass
        import checkers
        r = checkers.ref()  # the reference's own code (clang, libomp)
        if r is not None:
            r.ref_set_num_threads(32)
    yield


@pytest.fixture(scope="session")
def ref_lib():
    import checkers
    lib = checkers.ref()
    if lib is None:
        pytest.skip("oracle/_ref/libansel_ref.so not built (needs /root/reference)")
    return lib


@pytest.fixture(scope="session")
def oracle_lib():
    import checkers
    lib = checkers.oracle()
    assert lib is not None, "oracle/liboracle.so missing: run `python -c 'import __graft_entry__ as g; g.build()'`"
    return lib
