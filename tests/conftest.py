import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

GOLDEN = os.path.join(ROOT, "tests", "golden")
SCENES = os.path.join(GOLDEN, "scenes")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def scene_path(name):
    return os.path.join(SCENES, name + ".crtscene")


@pytest.fixture(scope="session")
def gpu_ctx():
    import cudaraytracing_amd as crt
    ctx = crt.Context(0)
    yield ctx
    ctx.close()
