import sys
from pathlib import Path

import pytest

REPO = Path(__file__).resolve().parent.parent
if str(REPO) not in sys.path:
    sys.path.insert(0, str(REPO))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a HIP device (runs on the MI355X box)")


@pytest.fixture(scope="session")
def golden_dir() -> Path:
    return REPO / "tests" / "golden"
