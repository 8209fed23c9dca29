import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "wfa-gpu_amd", "bindings"))
sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session", autouse=True)
def _build_checkers():
    """The oracle is test infrastructure: build it once per session (gcc only)."""
    import oracle_lib
    if not os.path.exists(oracle_lib.ORACLE_SO):
        oracle_lib.build()
    yield


@pytest.fixture(autouse=True)
def _default_launch_config(request):
    """GPU tests may reconfigure launch_alignments* (virtual devices, tuning hooks): back to the defaults afterwards,
    whatever the test's outcome."""
    yield
    if request.node.get_closest_marker("gpu") is not None:
        import wfagpu
        if wfagpu._lib is not None:
            wfagpu.configure_launch()
