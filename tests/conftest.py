import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def vpt():
    return importlib.import_module("vulkan-path-tracer_amd")


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle_py
    oracle_py.lib()
    return oracle_py


@pytest.fixture(scope="session")
def scenes(vpt):
    cache = {}

    def get(name):
        if name not in cache:
            cache[name] = vpt.scenes.Scene.load(os.path.join(GOLDEN, name + ".npz"))
        return cache[name]
    return get
