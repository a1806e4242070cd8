import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


# Memory checking of the GPU tier (zen_amd/csrc/memguard.h; the reference: cuda-memcheck, libzen/CMakeLists.txt:56-73): every
# allocation of the library -- the tests' DeviceBuffers and IOGPU buffers, the engines' own rings and rows -- carries 4 KB
# red zones and starts as NaNs; a fixture below verifies the zones after every GPU test.  Child processes (the C++ host
# tests, the CLI, bench.py) inherit the setting and exit with status 86 if a zone was found overwritten.
os.environ.setdefault("ZEN_HIP_REDZONE", "4096")
os.environ.setdefault("ZEN_HIP_POISON", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "memcheck_expected: the test provokes a red-zone / bounds finding on purpose")


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


@pytest.fixture(autouse=True)
def memcheck_after_gpu_test(request):
    """Red zones (and, on a -DZEN_HIP_BOUNDS build, the recorded out-of-bounds accesses) checked after every GPU test."""
    if request.node.get_closest_marker("gpu") is None:
        yield
        return
    import zen_amd
    try:
        before = zen_amd.memcheck()
    except zen_amd.ZenHipError as e:      # no GPU here: a gpu-marked test is skipped in its own name, not an error of this fixture
        pytest.skip("GPU test on a box without a usable GPU: %s" % e)
    yield
    after = zen_amd.memcheck()
    if request.node.get_closest_marker("memcheck_expected") is not None:
        return              # (the tests that break a store on purpose look at the report themselves)
    assert after["redzone_bytes"] >= 4096 or os.environ.get("ZEN_HIP_REDZONE") == "0", "the GPU tier runs with red zones"
    assert after["corrupt_words"] == before["corrupt_words"], after["first_message"]
    assert after["bounds_violations"] == before["bounds_violations"], after["first_message"]
