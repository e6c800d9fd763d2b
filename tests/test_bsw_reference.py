"""Seam S1 pinned against the REFERENCE's own kernels: oracle/_ref/refdump.<isa> bsw runs explicit (h0, query, target) pairs through
BandedPairWiseSW::getScores8 / getScores16 / scalarBandedSWAWrapper (each pair filed under the kernel sortPairsLenExt would file it under)
and dumps the six outputs (of the run in the order the reference runs a class in: sorted by target length).  The oracle's restatement (ksw_extend with the rule of the pair's kernel class) must give the same numbers --
including the `-O16` band wrap of the int8 / int16 wrappers (pairs with len2 * a + end_bonus - o < 0: SURVEY.md App. A #15), which until
round 3 had only been compared oracle-vs-device.  gpu: bm2_bsw against the same dump."""
import os
import subprocess

import numpy as np
import pytest

import bm2
from helpers import pack_pairs, random_pairs, ref_binary
from tools import oracle
import helpers  # noqa: E402

CASES = [  # (name, scoring, w, end_bonus, pairs)
    ("default", dict(), 100, 5, lambda: random_pairs(71, 1500, max_len=150, h0_max=150) + random_pairs(72, 40, max_len=900, h0_max=500)),
    ("intractg_wrap", dict(b=9, o_del=16, o_ins=16), 100, 5, lambda: random_pairs(73, 2500, max_len=30, h0_max=100) + random_pairs(74, 300, max_len=140, h0_max=120)),
    ("ont2d", dict(a=1, b=1, o_del=1, e_del=1, o_ins=1, e_ins=1), 100, 0, lambda: random_pairs(75, 400, max_len=600, h0_max=300)),
    ("small_band", dict(), 7, 5, lambda: random_pairs(76, 1200, max_len=200, h0_max=150)),
]


def _args(kw):
    o = oracle.default_opt(**kw)
    return ["-A", str(o.a), "-B", str(o.b), "-O", "%d,%d" % (o.o_del, o.o_ins), "-E", "%d,%d" % (o.e_del, o.e_ins), "-d", str(o.zdrop)]


def _reference(tmp_path, name, kw, w, end_bonus, triples):
    exe = ref_binary("refdump")
    if exe is None:
        helpers.no_checker("oracle/_ref not built (make -C oracle ref)")
    fn, out = str(tmp_path / (name + ".txt")), str(tmp_path / (name + ".bin"))
    with open(fn, "w") as f:
        for q, t, h0 in triples:
            f.write("%d %s %s\n" % (h0, "".join("ACGTN"[c] for c in q), "".join("ACGTN"[c] for c in t)))
    p = subprocess.run([exe] + _args(kw) + ["bsw", str(w), str(end_bonus), fn, out], stderr=subprocess.PIPE, text=True)
    assert p.returncode == 0, p.stderr[-400:]
    return np.fromfile(out, np.int32).reshape(-1, 8)


def _same(got, ref_row):
    """The six outputs; gtle only where it means something.  While gscore <= 0 the row of the best end-to-end score is whatever row last
    reached the query's end with H = 0, the caller never reads it (bwamem.cpp:2504-2511 takes qle / tle then), and the reference's vector
    kernels -- which keep stepping a finished pair while its SIMD neighbours run -- report a later row than its scalar kernel does."""
    r = tuple(int(x) for x in ref_row[:6])
    if r[4] <= 0:
        return got[:3] + got[4:] == r[:3] + r[4:]
    return got == r


@pytest.mark.parametrize("name,kw,w,end_bonus,make", CASES, ids=[c[0] for c in CASES])
def test_oracle_extension_equals_the_reference_kernels(tmp_path, name, kw, w, end_bonus, make):
    triples = make()
    ref = _reference(tmp_path, name, kw, w, end_bonus, triples)
    assert len(ref) == len(triples)
    o = oracle.default_opt(**kw)
    bad, wrap = [], 0
    for i, (q, t, h0) in enumerate(triples):
        exp = oracle.ksw_extend(q, t, o, w, end_bonus, h0)           # (score, qle, tle, gtle, gscore, max_off)
        if not _same(exp, ref[i]):
            bad.append((i, int(ref[i][6]), exp, tuple(int(x) for x in ref[i][:6])))
        wrap += len(q) * o.a + end_bonus - o.o_ins < 0 and int(ref[i][6]) != 32
    assert not bad, "%d of %d differ; first: pair %d class %d oracle %s reference %s" % (len(bad), len(triples), *bad[0])
    if name == "intractg_wrap":
        assert wrap > 300, wrap                                      # the wrapping pairs are really in the sample
    assert {8, 16} <= set(int(x) for x in ref[:, 6]) or name == "intractg_wrap"


@pytest.mark.gpu
@pytest.mark.parametrize("name,kw,w,end_bonus,make", CASES, ids=[c[0] for c in CASES])
def test_device_extension_equals_the_reference_kernels(gpu_ctx_factory, tmp_path, name, kw, w, end_bonus, make):
    triples = make()
    ref = _reference(tmp_path, name, kw, w, end_bonus, triples)
    pairs, refb, qerb = pack_pairs(bm2, triples)
    got = gpu_ctx_factory().bsw(pairs, refb, qerb, w, bm2.sw_params(bm2.default_opt(**kw), end_bonus))
    bad = [i for i in range(len(triples)) if not _same(tuple(int(got[i][f]) for f in ("score", "qle", "tle", "gtle", "gscore", "max_off")), ref[i])]
    assert not bad, "%d of %d differ; first: pair %d class %d" % (len(bad), len(triples), bad[0], int(ref[bad[0]][6]))
