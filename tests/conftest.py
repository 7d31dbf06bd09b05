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


# The library's own kernel watchdog (include/hnsw_gpu.h): a search launch that has been running for two minutes is asked to
# end through its workspace's abort word, so a hung kernel ends by itself and its test FAILS (wrong or missing answers)
# instead of keeping the device until somebody's limit kills the process.  Read once, at the library's first workspace.
os.environ.setdefault("HNSW_GPU_WATCHDOG_S", "120")

# The library resolves its knobs once and never reads the environment on a call path (hnsw_gpu_config_set is how a host changes
# one later).  The tests flip kernel forms by changing HNSW_GPU_* variables inside one process: with this set, pg_embedding_amd
# forwards such changes to the library before every launch (pg_embedding_amd/_lib.py, sync_env); child processes inherit it.
os.environ.setdefault("PGEMB_ENV_SYNC", "1")


def pytest_collection_modifyitems(config, items):
    """A device test that never returns must end the run with a failure, not hang it: pytest-timeout's watchdog thread
    ends the process after 10 minutes in one test (the slowest full-size test takes about one).  The device tier does
    not run without it: a missing plugin is a collection error, not a silent no-op."""
    gpu_items = [it for it in items if it.get_closest_marker("gpu")]
    if not gpu_items:
        return
    try:
        import pytest_timeout  # noqa: F401
    except Exception as e:
        deselected_all = config.getoption("-m") and "not gpu" in config.getoption("-m")
        if deselected_all:
            return                                            # the CPU tier collects them only to deselect them
        raise pytest.UsageError(f"the device tests need the pytest-timeout plugin (a hung kernel must fail, not hang): {e}")
    limit = int(os.environ.get("PGEMB_TEST_TIMEOUT", "600"))        # (a device session that hunts a hang sets it lower)
    for it in gpu_items:
        if not it.get_closest_marker("timeout"):
            it.add_marker(pytest.mark.timeout(limit, method="thread"))


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
