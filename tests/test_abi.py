"""The C-ABI library loads and exports every symbol include/bm2.h declares (no compute, no GPU)."""
import ctypes
import os
import re

import pytest

import bm2
from conftest import ROOT


def _declared():
    hdr = open(os.path.join(ROOT, "include", "bm2.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(bm2_[a-z0-9_]+)\s*\(", hdr)))


def test_header_and_binding_agree():
    assert _declared() == sorted(bm2.EXPORTS)


def test_library_exports_every_declared_symbol():
    assert os.path.exists(bm2.LIB_PATH), "libbm2.so not built: run __graft_entry__.build()"
    lib = ctypes.CDLL(bm2.LIB_PATH)
    for name in _declared():
        assert hasattr(lib, name), name


def test_record_sizes_match_reference_structs():
    # SMEM 40 B (FMI_search.h:75-83), SeqPair 56 B (bandedSWA.h:90-99) -- SURVEY.md App. B
    assert bm2.SMEM_DT.itemsize == 40
    assert bm2.SEQPAIR_DT.itemsize == 56
    assert ctypes.sizeof(bm2.Opt) == 120


def test_opt_defaults_are_mem_opt_init():
    o = bm2.default_opt()       # bwamem.cpp:107-143
    assert (o.a, o.b, o.o_del, o.e_del, o.o_ins, o.e_ins) == (1, 4, 6, 1, 6, 1)
    assert (o.w, o.zdrop, o.pen_clip5, o.pen_clip3) == (100, 100, 5, 5)
    assert (o.min_seed_len, o.split_width, o.max_occ, o.max_mem_intv) == (19, 10, 500, 20)
    assert list(o.mat)[:6] == [1, -4, -4, -4, -1, -4] and list(o.mat)[20:] == [-1] * 5


def test_no_device_fails_loudly():
    if bm2.lib().bm2_device_count() > 0:
        pytest.skip("a HIP device is visible here: the no-device error path cannot be reached")
    try:
        bm2.Context(0)
    except bm2.Bm2Error as e:
        assert "no HIP device" in str(e)
    else:
        raise AssertionError("bm2_create must fail without a GPU: there is no CPU fallback")
