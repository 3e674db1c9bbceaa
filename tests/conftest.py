import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


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
