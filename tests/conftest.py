import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `pytest -m gpu` on the GPU box)")


@pytest.fixture(scope="session")
def emu_lib():
    """CPU model of the kernels (tests/emu): same sources as the product, compiled with g++ -DMI355_EMU.
    Test infrastructure only — checks kernel/host logic without a GPU."""
    from mimic3_amd import build
    from mimic3_amd._native import NativeLibrary

    return NativeLibrary(build.build_emu())


@pytest.fixture(scope="session")
def gpu_lib():
    """The product library on the real device; fails loudly if it is missing."""
    from mimic3_amd._native import default_library

    return default_library()


@pytest.fixture(scope="session")
def gpu_hooks():
    """libmi355vits_hooks.so on the real device: the PRODUCT's object files + the kernel unit-test / probe hooks of
    include/mi355vits_lab.h (the product library exports include/mi355vits.h only).  Test infrastructure."""
    from mimic3_amd._native import hooks_library

    return hooks_library()


@pytest.fixture(scope="session")
def lab_lib():
    """The LAB build of the HIP library (-DMI355_LAB: kernel-choice switches for A/B tests) on the real device; built in-tree by
    ``__graft_entry__.build()`` / ``python -m mimic3_amd.build lab``.  Test infrastructure: the product never opens it."""
    from mimic3_amd._native import NativeLibrary

    path = os.path.join(ROOT, "mimic3_amd", "csrc", "libmi355vits_lab.so")
    if not os.path.exists(path):
        pytest.skip("lab build of the library not present (python -m mimic3_amd.build lab)")
    return NativeLibrary(path)
