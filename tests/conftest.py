import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import _vitpkg  # noqa: E402


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def O():
    """The CPU oracle (test infrastructure)."""
    mod = _vitpkg.load_oracle()
    mod.build()
    return mod


@pytest.fixture(scope="session")
def V():
    """The product package; libviterbi.so must already be built (never silently rebuilt on a GPU box)."""
    mod = _vitpkg.load_package()
    if not os.path.exists(mod.LIB_PATH):
        mod.build()
    # The library's default comparator is the MASM decoders' `>= 150` (what an installed viterbi.dll runs; asserted in
    # tests/test_abi.py).  The oracle's default, and what most tests compare with, is the C decoders' `> 150`: the
    # session runs in that mode, and the tests of the comparator select each mode explicitly.
    mod.set_renorm_ge(0)
    return mod


@pytest.fixture(scope="session")
def torch_cuda(V):
    import torch
    if not torch.cuda.is_available():
        pytest.fail("gpu test selected but torch sees no GPU")
    assert V.device_count() >= 1, "libviterbi.so found no gfx950 device: " + V.last_error()
    V.initialize()
    return torch
