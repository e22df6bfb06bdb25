"""Loader for the hyphen-named product package and the oracle binding."""
import importlib.util, os, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG_DIR = os.path.join(ROOT, "structure-slam-pointline_amd")


def _load(name, path):
    if name in sys.modules:
        return sys.modules[name]
    spec = importlib.util.spec_from_file_location(name, path)
    m = importlib.util.module_from_spec(spec)
    sys.modules[name] = m
    spec.loader.exec_module(m)
    return m


def frontend():
    return _load("sslam_frontend", os.path.join(PKG_DIR, "frontend.py"))


def builder():
    return _load("sslam_build", os.path.join(PKG_DIR, "build.py"))
