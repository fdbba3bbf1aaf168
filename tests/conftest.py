import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return ROOT / "tests" / "golden"


@pytest.fixture
def tune():
    """set run-time knobs of libyolov3_hip.so (y3_tune_set, include/yolov3_hip.h) for one test; defaults are restored afterwards"""
    from yolov3_amd import ops

    yield ops.tune_set
    ops.tune_reset()
