import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


# torch bundles its own HIP runtime: when a test process is going to use both torch and libzolt_gpu.so, torch has to be
# imported FIRST (afterwards it fails with "No HIP GPUs are available"). Test modules that need torch import it at module level,
# which pytest's collection runs before any test body; importing it here as well makes that independent of which files are selected.
try:
    import torch  # noqa: F401
except Exception:  # pragma: no cover - torch is optional for the CPU suite
    torch = None


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="session", autouse=True)
def _pool_debug_report():
    """ZG_POOL_DEBUG=1 python -m pytest tests -m gpu: the whole suite runs on poisoned / verified pool blocks (csrc/runtime.hip); a block
    written after its free fails the call that would have received it, and the session ends with the counters (zero hits required)."""
    yield
    if os.environ.get("ZG_POOL_DEBUG"):
        from zolt_amd import lib
        st = lib.pool_debug_stats()
        print(f"\nZG_POOL_DEBUG stats: {st}")
        assert st["hits"] == 0, st
