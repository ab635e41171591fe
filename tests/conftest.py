import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def solr():
    mod = importlib.import_module("sol-r_amd")
    if not (os.path.exists(mod.HIP_LIB) and os.path.exists(mod.HOST_LIB)):
        mod.build()
    return mod


@pytest.fixture(scope="session")
def oracle():
    from oracle import loader
    loader.lib()
    return loader


@pytest.fixture(scope="session")
def have_gpu(solr):
    return solr.hip_lib().solr_hip_device_count() > 0
