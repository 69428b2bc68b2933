import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run through gpurun)')


def pytest_collection_modifyitems(config, items):
    # `-m gpu` on a box without a GPU should fail loudly, not silently skip: the GPU tests
    # assert torch.cuda.is_available() themselves.
    pass


@pytest.fixture(scope='session')
def d2p_lib():
    from demo2program_amd import build, lib
    build.build_library()
    return lib.load()
