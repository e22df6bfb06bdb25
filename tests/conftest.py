import os, sys
import pytest
try:                      # torch first: it ships its own HIP runtime, and a process that initialised the system one first (through the library under
    import torch          # test) finds no device from torch afterwards ("No HIP GPUs are available"); the harness modules import torch at their top too
except Exception:         # pragma: no cover
    torch = None

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    import oracle_lib
    return oracle_lib.Oracle()


@pytest.fixture(scope="session")
def fe():
    import pkg
    return pkg.frontend()


@pytest.fixture(scope="session")
def ctx(fe):
    c = fe.Context(0)
    yield c
    c.close()
