import importlib
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pkg():
    return importlib.import_module("3dgs_monocular_depth_init_amd")


@pytest.fixture(scope="session")
def rendering():
    return importlib.import_module("3dgs_monocular_depth_init_amd.rendering")


def pytest_sessionfinish(session, exitstatus):
    from tests import parity_log
    parity_log.dump()
