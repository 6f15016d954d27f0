import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def ora32():
    from oracle.oracle import Oracle
    return Oracle("f32")


@pytest.fixture(scope="session")
def ora64():
    from oracle.oracle import Oracle
    return Oracle("f64")


def _backend_params():
    return ["emu", pytest.param("hip", marks=pytest.mark.gpu)]


@pytest.fixture(scope="session", params=_backend_params())
def backend(request):
    from common import Backend
    return Backend(request.param)
