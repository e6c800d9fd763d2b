"""Device sources executed on the host (tools/emu: fake hip_runtime.h, one OS thread per GPU thread, a rendezvous per wave-level
primitive) against the oracle.  k_bsw_pairs and the wave-per-task DP of bsw_dev.h -- DPP max-scans, ballots, readlanes, LDS rings --
are GPU-verified code: that they give the oracle's answers here too is the check of the EMULATOR, which is what lets new kernels
(notes/*.patch) be debugged before they meet a GPU.  Small inputs: a primitive costs ~100 us here."""
import os
import subprocess

import numpy as np
import pytest

from helpers import random_pairs
from tools import oracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EMU = os.path.join(ROOT, "tools", "emu")
CSRC = os.path.join(ROOT, "bwa-mem2_amd", "csrc")


def build(tmp, src_name, harness, macro):
    """The emu build of one device source: `extern __shared__` -> a harness-defined array, then g++ against the fake HIP header."""
    src = os.path.join(tmp, src_name + ".cpp")
    with open(os.path.join(CSRC, src_name)) as f, open(src, "w") as g:
        g.write(f.read().replace("extern __shared__", "EMU_EXTERN_SHARED"))
    exe = os.path.join(tmp, harness.replace(".cpp", ""))
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-pthread", "-w", "-I", os.path.join(EMU, "fakehip"), "-I", CSRC,
                           "-D%s=\"%s\"" % (macro, src), os.path.join(EMU, harness), os.path.join(EMU, "emu_runtime.cpp"), "-o", exe])
    return exe


@pytest.fixture(scope="module")
def bsw_emu(tmp_path_factory):
    return build(str(tmp_path_factory.mktemp("emu")), "bsw.hip", "bsw_emu.cpp", "BSW_SRC")


@pytest.mark.parametrize("seed,n,max_len,w", [(1, 10, 40, 100), (2, 2, 300, 20)])
def test_k_bsw_pairs_on_the_emulator(bsw_emu, tmp_path, seed, n, max_len, w):
    tr = random_pairs(seed, n, max_len=max_len, h0_max=100)
    opt = oracle.default_opt()
    pf, of = str(tmp_path / "pairs.txt"), str(tmp_path / "out.bin")
    with open(pf, "w") as f:
        for q, t, h0 in tr:
            f.write("%d %s %s\n" % (h0, "".join("ACGTN"[c] for c in q), "".join("ACGTN"[c] for c in t)))
    subprocess.check_call([bsw_emu, pf, of], env=dict(os.environ, W=str(w)), timeout=900)
    got = np.fromfile(of, "<i4").reshape(-1, 6)
    for i, (q, t, h0) in enumerate(tr):
        assert tuple(got[i]) == tuple(oracle.ksw_extend(q, t, opt, w, 5, h0)), (i, len(q), len(t), h0)
