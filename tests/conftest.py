import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")
    # The engine builds its order-free node lists when a scene has stayed for a frame (a host that uploads the scene
    # again for every frame never pays for them).  Most tests render one frame of a scene: here the lists are built
    # with the first frame, so that every parity test also holds the order-free walks to the oracle;
    # tests/test_gpu_parity.py::test_order_free_lists_arrive_with_the_second_frame covers the default.
    os.environ.setdefault("SOLR_HIP_FREE_AFTER", "1")


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
