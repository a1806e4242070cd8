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
    from oracle import oracle as o
    o.lib()
    return o


@pytest.fixture(scope="session", autouse=True)
def built_artifacts():
    """The shared libraries are git-ignored build products.  A fresh checkout gets them built once (hipcc
    cross-compiles gfx950 without a GPU); this is the test harness building the product, not a fallback:
    zen_amd.load() itself still raises when libzen_hip.so is absent."""
    from zen_amd import build
    so = os.path.join(ROOT, "zen_amd", "libzen_hip.so")
    host = os.path.join(ROOT, "zen_amd", "libzen.so")
    if not os.path.exists(so):
        build.build()
    if not os.path.exists(host):
        from oracle import oracle as o
        o.build()
        build.build_host()
