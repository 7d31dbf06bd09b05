import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # The oracle must exist before collection: test modules parametrise on oracle.have_ref().
    import oracle
    oracle.build_oracle()


def pytest_collection_modifyitems(config, items):
    """A device test that never returns (a kernel that waits for something that cannot happen keeps its caller polling)
    must end the run with a failure, not hang it: pytest-timeout's watchdog thread ends the process after 15 minutes in
    one test (the slowest full-size test takes about one)."""
    try:
        import pytest_timeout  # noqa: F401
    except Exception:
        return
    for it in items:
        if it.get_closest_marker("gpu") and not it.get_closest_marker("timeout"):
            it.add_marker(pytest.mark.timeout(900, method="thread"))


@pytest.fixture(scope="session", autouse=True)
def _built():
    """Make sure the oracle (gcc) and the product libraries (hipcc, cross-compiles) exist."""
    import oracle
    oracle.build_oracle()
    from pg_embedding_amd import build as b
    b.build()


@pytest.fixture(scope="session")
def gpu_count():
    from pg_embedding_amd._lib import gpu_lib
    return gpu_lib().hnsw_gpu_device_count()
