"""The device source of the mate-rescue SW (bwa-mem2_amd/csrc/matesw_dev.h) executed on the host by tools/emu (one thread per
lane of a 16-lane row, a barrier at every cross-lane primitive) against the host kernel that is pinned to the reference's
ksw_align2.  Small tasks only: the emulation spends tens of microseconds per primitive.  What this cannot check -- the hardware
meaning of DPP row_shr:1 / ballot / __shfl -- is what tests/test_ksw_align2_gpu.py is for."""
import os
import subprocess

import numpy as np
import pytest

import bm2
from test_ksw_align2 import KSW_XBYTE, KSW_XSTART, KSW_XSUBO

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def emu(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("emu") / "matesw_emu")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-pthread", os.path.join(ROOT, "tools", "emu", "matesw_emu.cpp"), "-o", exe])
    return exe


@pytest.mark.parametrize("reg", [0, 1], ids=["lds_rows", "register_rows"])
@pytest.mark.parametrize("kw", [{}, dict(a=2, b=5, o_del=7, o_ins=8, e_del=2, e_ins=1)])
def test_device_source_on_the_lane_emulator(emu, tmp_path, kw, reg):
    opt = bm2.default_opt(**kw)
    rng = np.random.default_rng(3 + len(kw))
    pairs, xtra = [], []
    for i in range(14):
        ql = int(rng.integers(1, 70 if i % 4 else 161)); tl = int(rng.integers(1, 160))       # (up to ten stripe segments of the byte kernel: what the register pass takes)
        t = rng.integers(0, 4, tl, dtype=np.uint8)
        q = rng.integers(0, 4, ql, dtype=np.uint8)
        if i % 3 and tl > ql:                                    # plant the query with a few differences
            s = int(rng.integers(0, tl - ql + 1))
            q = t[s:s + ql].copy()
            m = rng.random(ql) < 0.08
            q[m] = (q[m] + 1) % 4
        pairs.append((q, t))
        x = (5 * opt.a) | (KSW_XSUBO if i % 5 else 0) | (KSW_XSTART if i % 7 else 0) | (KSW_XBYTE if i % 2 else 0)
        xtra.append(x)
    exp = bm2.ksw_align2(pairs, xtra, opt)
    pf, of = str(tmp_path / "pairs.txt"), str(tmp_path / "out.bin")
    with open(pf, "w") as f:
        for (q, t), x in zip(pairs, xtra):
            f.write("%d %s %s\n" % (x, "".join("ACGTN"[c] for c in q), "".join("ACGTN"[c] for c in t)))
    if reg:
        os.environ["EMU_KSW_REG"] = "1"
    else:
        os.environ.pop("EMU_KSW_REG", None)
    env = dict(os.environ, A=str(opt.a), B=str(opt.b), O_DEL=str(opt.o_del), E_DEL=str(opt.e_del), O_INS=str(opt.o_ins), E_INS=str(opt.e_ins))
    subprocess.check_call([emu, pf, of], env=env, timeout=600)
    got = np.fromfile(of, "<i4").reshape(-1, 7)
    assert (exp == got).all(), (exp.tolist(), got.tolist())
