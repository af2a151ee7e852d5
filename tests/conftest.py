import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def srt():
    lib_path = os.path.join(ROOT, "cuda-spectral-ray-tracer_amd", "libsrt_hip.so")
    if not os.path.exists(lib_path):
        import __graft_entry__
        __graft_entry__.build()
    return importlib.import_module("cuda-spectral-ray-tracer_amd")


@pytest.fixture(scope="session")
def orc():
    import oracle_binding
    oracle_binding.lib()
    return oracle_binding


@pytest.fixture(scope="session")
def gpu(srt):
    """One device context for the whole GPU session (tests run in one process)."""
    r = srt.Renderer(0)      # raises loudly if there is no GPU: no CPU fallback exists
    r.set_gather_planes(9)   # parity suite: the kernels also write the unquantised sRGB / XYZ planes (default 3: the framebuffer only)
    yield r
    r.close()
