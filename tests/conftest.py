import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session", autouse=True)
def _native_built():
    """Build the in-tree native libraries once (hipcc cross-compiles without a GPU)."""
    from flooder_amd import build

    try:
        build.build_all()
    except Exception as exc:  # pragma: no cover
        print(f"native build failed: {exc}", file=sys.stderr)
    yield
