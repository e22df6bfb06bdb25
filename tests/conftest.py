import os, sys
import pytest

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
