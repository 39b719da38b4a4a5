import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle.pyoracle import Oracle
    return Oracle()


@pytest.fixture(scope="session")
def reference():
    from oracle import pyoracle
    if not pyoracle.have_reference():
        if os.path.isdir(pyoracle.REF_SRC):
            pyoracle.build(ref=True)
        else:
            pytest.skip("compiled reference (oracle/_ref) not available here")
    return pyoracle.Reference()


@pytest.fixture(scope="session")
def gpu_lib():
    from pangenomenem_amd import build, engine
    build.build()
    lib = engine.load_library()
    if engine.device_count() <= 0:
        pytest.fail("no HIP device visible: the gpu tests must run on the GPU box")
    return lib
