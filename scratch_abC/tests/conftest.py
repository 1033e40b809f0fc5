import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture
def env(request):
    """(device, backend name).  Tests marked gpu use the real gfx950 library on cuda:0; all others run the SAME kernel
    sources compiled for the CPU emulator (tests/emu) on CPU tensors -- test infrastructure only."""
    import torch
    from learningbycheating_amd import _lib
    if request.node.get_closest_marker("gpu") is not None:
        if not torch.cuda.is_available():
            pytest.skip("no GPU")
        _lib._inject_for_tests(None)
        lib = _lib.load()
        assert lib.lbc_backend().decode() == "hip-gfx950", "GPU tests must run the HIP library"
        yield torch.device("cuda", 0), "hip-gfx950"
    else:
        from tests import emu
        emu.activate()
        yield torch.device("cpu"), "emu-cpu"
        emu.deactivate()


@pytest.fixture
def lbc_config(env):
    """set runtime options of the loaded library (lbc_config_set; names = the LBC_* variables) for one test, restored after"""
    from learningbycheating_amd import _lib
    saved = {}

    def set_opt(name, value):
        if name not in saved:
            saved[name] = _lib.config_get(name)
        _lib.config_set(name, value)
    yield set_opt
    for k, v in saved.items():
        _lib.config_set(k, v)
