"""The kbtree restatement of chain.hip (bwamem.cpp:906-951 through kbtree.h's kb_putp / kb_intervalp / __kb_traverse with t = 5) in all the forms the chaining
kernels run it in -- nodes probed in memory, visited through registers, internal nodes in a second pool (k_chain_serial's LDS pool, also when it overflows),
second-pool nodes probed in place -- compiled for the host and driven by one thread over the same random insertion sequences, most of them rich in EQUAL keys:
the in-order key sequence and every lower-bound look-up must agree between the forms at all times.  (The kernels themselves: tests/test_pipeline_gpu.py on a GPU;
long reads are beyond what the thread-per-lane emulator can run in minutes.)"""
import ctypes
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def bt_lib(emu_lib, tmp_path_factory):
    sys.path.insert(0, os.path.join(ROOT, "tools", "emu"))
    import build_emu
    d = str(tmp_path_factory.mktemp("btpool"))
    src = os.path.join(d, "chain_emu.cpp")
    with open(os.path.join(ROOT, "bwa-mem2_amd", "csrc", "chain.hip")) as g:
        open(src, "w").write(build_emu.rewrite(g.read()))
    so = os.path.join(d, "libbtpool.so")
    cxx = "/opt/rocm/lib/llvm/bin/clang++" if os.path.exists("/opt/rocm/lib/llvm/bin/clang++") else "g++"
    emu_dir = os.path.dirname(emu_lib)
    subprocess.check_call([cxx, "-O1", "-std=c++17", "-pthread", "-w", "-fPIC", "-shared", "-I", os.path.join(ROOT, "tools", "emu", "fakehip"),
                           "-I", os.path.join(ROOT, "bwa-mem2_amd", "csrc"), '-DCHAIN_EMU_CPP="%s"' % src, os.path.join(ROOT, "tests", "btree_pool", "driver.cpp"),
                           "-L", emu_dir, "-l:" + os.path.basename(emu_lib), "-Wl,-rpath," + emu_dir, "-o", so])
    lib = ctypes.CDLL(so)
    lib.bt_pool_check.restype = ctypes.c_int
    lib.bt_pool_check.argtypes = [ctypes.c_uint, ctypes.c_int, ctypes.c_longlong, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_longlong)]
    return lib


# (keys, span of the key values, capacity of the second pool): spans far below the key count make equal keys the rule; pool capacities 0 (nothing fits: every node in
# the first pool), 1-3 (the root moves in, its successors overflow), enough for all internal nodes
@pytest.mark.parametrize("n,span,l_cap", [(40, 7, 0), (400, 50, 1), (400, 50, 3), (3000, 300, 8), (3000, 1 << 40, 40), (20000, 4000, 25), (20000, 1 << 33, 2000),
                                          (60000, 9000, 960), (60000, 1 << 34, 960)])
def test_every_form_of_the_tree_is_the_same_tree(bt_lib, n, span, l_cap):
    detail = (ctypes.c_longlong * 4)()
    for seed in range(3):
        rc = bt_lib.bt_pool_check(seed * 7919 + n, n, span, l_cap, 2 if n <= 20000 else 1, detail)
        assert rc == 0, "check %d failed (seed %d): %s" % (rc, seed, list(detail))
        n_l, n_glob, n_all = detail[0], detail[1], detail[2]
        assert n_l <= max(l_cap, 0) and n_l + n_glob == n_all
        if l_cap >= 960:
            assert n_l > 100, "the second pool was hardly used: %d nodes" % n_l
