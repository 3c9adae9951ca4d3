import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box)")


@pytest.fixture(scope="session")
def tor():
    """The product package (directory name has a hyphen -> importlib)."""
    mod = importlib.import_module("trace-of-radiance_amd")
    if not os.path.exists(mod.LIB_PATH):
        mod.build()
    return mod


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.build()
    return O


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="session")
def ref_scene(oracle):
    objs, draws = oracle.random_scene(0xFACADE)
    return objs, draws


@pytest.fixture(scope="session")
def ref_camera(oracle):
    return oracle.camera()
