import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "bwa-mem2_amd"))

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


# BM2_EMU_LIB=<libbm2_emu.so built by tools/emu/build_emu.py>: the `gpu` tests run against the host emulator of the device sources
# (a logic check without a GPU; sizes are what they are, so pick tests with -k).
if os.environ.get("BM2_EMU_LIB"):
    import bm2 as _bm2
    _bm2.LIB_PATH = os.environ["BM2_EMU_LIB"]


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def gpu_ctx_factory():
    """Creates bm2 contexts on cuda:0; fails loudly (no CPU fallback) if the library or device is missing."""
    import bm2
    made = []

    def make(index_prefix=None):
        c = bm2.Context(0, index_prefix)
        made.append(c)
        return c
    yield make
    for c in made:
        c.close()
