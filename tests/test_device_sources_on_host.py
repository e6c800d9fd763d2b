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
    """One device source + a harness with its own main(), against the fake HIP header."""
    import sys
    sys.path.insert(0, EMU)
    import build_emu
    src = os.path.join(tmp, src_name + ".cpp")
    with open(os.path.join(CSRC, src_name)) as f, open(src, "w") as g:
        g.write(build_emu.rewrite(f.read()))
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


@pytest.fixture(scope="module")
def emu_lib(tmp_path_factory):
    import sys
    sys.path.insert(0, EMU)
    import build_emu
    return build_emu.build(str(tmp_path_factory.mktemp("emulib")))


def test_whole_device_pipeline_on_the_emulator(emu_lib, golden_dir):
    # every kernel of bm2_seed_chain_extend (seeding task kernels with their quad-cooperative Occ loads, SA lookup, chaining, the
    # lane-per-task extension rounds, the purge) executed by OS threads, against the oracle: 3 reads take about a minute
    script = r'''
import sys
sys.path.insert(0, %r); sys.path.insert(0, %r); sys.path.insert(0, %r)
import numpy as np, bm2
bm2.LIB_PATH = %r
from helpers import load_golden, regs_to_records
from tools import oracle
pre, enc, off, ln, d = load_golden(%r, "g20k_l76")
n = 3
ln = ln[:n]; off = off[:n]; enc = enc[:int(off[-1] + ln[-1])]
ctx = bm2.Context(0, pre)
regs, reg_off, st = ctx.seed_chain_extend(enc, off, ln, bm2.default_opt())
ix = oracle.Index(pre); exp = ix.run(enc, off, ln); ix.close()
assert st["n_smem"] == len(exp["SMEM"]) and st["n_sa"] == len(exp["SACOORD"]), st
assert regs_to_records(regs, reg_off).tobytes() == exp["REGPRG"].tobytes()
print("ok", len(regs))
''' % (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "bwa-mem2_amd"), emu_lib, golden_dir)
    p = subprocess.run(["python", "-c", script], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=1500)   # own process: bm2 binds one library
    assert p.returncode == 0 and p.stdout.startswith(b"ok"), p.stderr.decode()[-2000:]
