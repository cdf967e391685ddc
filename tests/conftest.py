import os
import sys

import pytest

try:                    # before librt_amd.so is loaded by any fixture: PyTorch brings its own copy of the HIP runtime, and a process in which the
    import torch        # library's copy came first leaves torch without a device ("No HIP GPUs are available"); bench.py keeps the same order
except ImportError:     # noqa: F401
    torch = None

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def golden_path(name):
    """A committed test fixture under tests/golden/ (oracle outputs, small decoder inputs).  The reference's own data assets that the
    workloads need at run time ship with the package: raytracinginrust_amd/assets/, `scenes.asset_path`."""
    return os.path.join(ROOT, "tests", "golden", name)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


@pytest.fixture(scope="session")
def obe():
    """The CPU oracle (test infrastructure) as a builder backend."""
    from oracle import orc
    return orc.load()


@pytest.fixture(scope="session")
def pbe():
    """The product library librt_amd.so; a missing library is an error, never a skip."""
    from raytracinginrust_amd import _lib
    return _lib.load()


@pytest.fixture(scope="session")
def earth():
    """The reference's own 1024x512 earth texture (raytracinginrust_amd/assets/earthmap.jpg = its earthmap.jpg, src/main.rs:491-495), decoded by
    the library's JPEG ingest -> (bytes, w, h)."""
    from raytracinginrust_amd import scenes
    return scenes.load_earthmap()


def build_scene(name, backend, earth=None):
    from raytracinginrust_amd import scenes
    if name == "cornell":
        return scenes.cornell_box(backend)
    if name == "random":
        return scenes.random_scene(backend, aspect_ratio=16.0 / 9.0)
    if name == "final":
        return scenes.final_scene(backend, *earth)
    if name == "teapot":
        return scenes.cornell_test(backend, scenes.asset_path("teapot.obj"), aspect_ratio=16.0 / 9.0)
    raise KeyError(name)
