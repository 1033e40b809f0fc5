"""TEST INFRASTRUCTURE ONLY: builds (on demand) and loads the CPU-emulated build of the HIP
kernel sources (see tests/emu/include/hip/hip_runtime.h) and injects it into the package loader."""
import ctypes
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
EMU_LIB = os.path.join(HERE, "liblbc_emu.so")


def build():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "learningbycheating_amd", "csrc"), "emu"])
    return EMU_LIB


def activate():
    from learningbycheating_amd import _lib
    build()
    return _lib._inject_for_tests(ctypes.CDLL(EMU_LIB))


def deactivate():
    from learningbycheating_amd import _lib
    _lib._inject_for_tests(None)
