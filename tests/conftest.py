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
def emu_lib(tmp_path_factory):
    """The device sources compiled for the host (tools/emu/build_emu.py): built ONCE per test session (a build takes ~20 s)."""
    sys.path.insert(0, os.path.join(ROOT, "tools", "emu"))
    import build_emu
    return build_emu.build(str(tmp_path_factory.mktemp("emulib")))


def emu_dir_as_libbm2(emu_lib, tmp_path):
    """A directory holding the emulator library under the name the bindings link against (for LD_LIBRARY_PATH)."""
    d = os.path.join(str(tmp_path), "emu")
    os.makedirs(d, exist_ok=True)
    os.symlink(emu_lib, os.path.join(d, "libbm2.so"))
    return d


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
