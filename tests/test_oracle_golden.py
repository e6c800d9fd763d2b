"""The plain-C restatement (oracle/bm2_oracle.c) against fixtures dumped from the COMPILED REFERENCE
(tools/make_golden.py -> oracle/refdump.cpp).  Every stage of the hot path, byte for byte.  No GPU."""
import numpy as np
import pytest

from helpers import ONT2D, chain_mask, first_diff, load_golden
from tools import oracle

STAGES = ["SMEM", "SACOORD", "SACNT", "CHN0", "SEED0", "CHN1", "SEED1", "REGRAW", "REGPRG"]


@pytest.mark.parametrize("name", ["g60k", "g20k_l76"])
def test_oracle_matches_reference_dump(golden_dir, name):
    pre, enc, off, ln, d = load_golden(golden_dir, name)
    ix = oracle.Index(pre)
    try:
        r = ix.run(enc, off, ln)
    finally:
        ix.close()
    for k in STAGES:
        exp, got = d[k], r[k]
        if k == "CHN0":
            exp, got = chain_mask(exp), chain_mask(got)
        assert len(exp) == len(got) and exp.tobytes() == got.tobytes(), "%s: %s" % (k, first_diff(exp, got))
    c = r["counters"]
    assert c["n_sa_lookup"] == len(d["SACOORD"]) and c["n_ext"] > 0 and c["n_sw_cells"] > 0


def test_oracle_seeding_only_and_empty(golden_dir):
    pre, enc, off, ln, d = load_golden(golden_dir, "g20k_l76")
    ix = oracle.Index(pre)
    try:
        r = ix.run(enc, off, ln, seeding_only=True)
        assert r["SMEM"].tobytes() == d["SMEM"].tobytes() and len(r["REGRAW"]) == 0
        e = ix.run(np.zeros(0, np.uint8), np.zeros(0, np.int64), np.zeros(0, np.int32))
        assert len(e["SMEM"]) == 0 and len(e["REGPRG"]) == 0
    finally:
        ix.close()


def test_ksw_extend_known_answers():
    # hand-checkable cases of ksw_extend2 (bandedSWA.cpp:116-237): perfect match, and no positive extension
    o = oracle.default_opt()
    q = np.array([0, 1, 2, 3, 0, 1, 2, 3], np.uint8)
    sc, qle, tle, gtle, gscore, max_off = oracle.ksw_extend(q, q, o, 100, 5, 10)
    assert (sc, qle, tle, gtle, gscore, max_off) == (18, 8, 8, 8, 18, 0)
    t = (q + 1) % 4
    sc, qle, tle, gtle, gscore, max_off = oracle.ksw_extend(q, t, o, 100, 5, 3)
    assert sc == 3 and qle == 0 and tle == 0


def test_oracle_long_reads_ont2d(golden_dir):
    # `-x ont2d`: mem_flt_chained_seeds / mem_seed_sw (local SW per short seed), min_chain_weight, kb-long extensions
    pre, enc, off, ln, d = load_golden(golden_dir, "g40k_ont")
    ix = oracle.Index(pre)
    try:
        r = ix.run(enc, off, ln, oracle.default_opt(**ONT2D))
    finally:
        ix.close()
    for k in STAGES:
        exp, got = d[k], r[k]
        if k == "CHN0":
            exp, got = chain_mask(exp), chain_mask(got)
        assert len(exp) == len(got) and exp.tobytes() == got.tobytes(), "%s: %s" % (k, first_diff(exp, got))
