"""S1 parity: bm2_bsw (HIP, one task per wavefront) vs the oracle's ksw_extend2 restatement, bit-exact on all six
outputs (score, qle, tle, gtle, gscore, max_off) -- the fields `xeonbsw -DMAXI` prints (test/main_banded.cpp)."""
import numpy as np
import pytest

import bm2
from helpers import pack_pairs, random_pairs
from tools import oracle

pytestmark = pytest.mark.gpu


def _check(ctx, triples, oopt, bopt, w, end_bonus):
    pairs, ref, qer = pack_pairs(bm2, triples)
    got = ctx.bsw(pairs, ref, qer, w, bm2.sw_params(bopt, end_bonus))
    bad = 0
    for i, (q, t, h0) in enumerate(triples):
        exp = oracle.ksw_extend(q, t, oopt, w, end_bonus, h0)
        g = tuple(int(got[i][f]) for f in ("score", "qle", "tle", "gtle", "gscore", "max_off"))
        if g != exp:
            bad += 1
            if bad <= 5:
                print("pair", i, "len", len(q), len(t), "h0", h0, "got", g, "exp", exp)
    assert bad == 0, "%d / %d pairs differ" % (bad, len(triples))


@pytest.mark.parametrize("w", [100, 200])
def test_bsw_default_scoring(gpu_ctx_factory, w):
    ctx = gpu_ctx_factory()
    tr = random_pairs(11 + w, 3000)
    _check(ctx, tr, oracle.default_opt(), bm2.default_opt(), w, 5)


def test_bsw_small_band_and_tiny(gpu_ctx_factory):
    ctx = gpu_ctx_factory()
    tr = random_pairs(5, 1500, max_len=40, h0_max=60)
    _check(ctx, tr, oracle.default_opt(), bm2.default_opt(), 7, 5)
    tr = random_pairs(6, 500, max_len=3, h0_max=20)
    _check(ctx, tr, oracle.default_opt(), bm2.default_opt(), 100, 5)


def test_bsw_ont2d_scoring_long(gpu_ctx_factory):
    ctx = gpu_ctx_factory()
    kw = dict(a=1, b=1, o_del=1, e_del=1, o_ins=1, e_ins=1, pen_clip5=0, pen_clip3=0)
    tr = random_pairs(21, 300, max_len=1500, h0_max=400)
    _check(ctx, tr, oracle.default_opt(**kw), bm2.default_opt(**kw), 100, 0)
    tr = random_pairs(22, 20, max_len=6000, h0_max=2000, long_tail=False)
    _check(ctx, tr, oracle.default_opt(**kw), bm2.default_opt(**kw), 200, 0)


def test_bsw_intractg_band_wrap(gpu_ctx_factory):
    # -x intractg (O=16, B=9): pairs with len2*a + L - O < 0 take the wrapping band of the int8/int16 wrappers
    ctx = gpu_ctx_factory()
    kw = dict(b=9, o_del=16, o_ins=16)
    tr = random_pairs(31, 2000, max_len=30, h0_max=100)
    _check(ctx, tr, oracle.default_opt(**kw), bm2.default_opt(**kw), 100, 5)


def test_bsw_zdrop_rule_of_each_kernel_class(gpu_ctx_factory):
    # e_del / e_ins != 1, zdrop <= 0, zdrop >= 128: the int8 / int16 kernels of the reference stop on a different rule than its
    # scalar kernel (ZSCORE8/16, bandedSWA.cpp:268-281, 309-322); pairs of all three classes, both sides of the 128 / 32768 limits
    ctx = gpu_ctx_factory()
    for seed, kw in ((41, dict(e_del=2, e_ins=3, zdrop=30)), (42, dict(zdrop=0)), (43, dict(zdrop=200)), (44, dict(zdrop=-5, e_ins=2)),
                     (45, dict(a=2, b=8, o_del=12, o_ins=12, e_del=2, e_ins=2, zdrop=200, pen_clip5=10, pen_clip3=10))):
        tr = random_pairs(seed, 1500, max_len=200, h0_max=150)
        _check(ctx, tr, oracle.default_opt(**kw), bm2.default_opt(**kw), 100, 5)
        tr = random_pairs(seed + 100, 40, max_len=1200, h0_max=600)
        _check(ctx, tr, oracle.default_opt(**kw), bm2.default_opt(**kw), 100, 0)


def test_bsw_empty_batch(gpu_ctx_factory):
    ctx = gpu_ctx_factory()
    out = ctx.bsw(np.zeros(0, bm2.SEQPAIR_DT), np.zeros(1, np.uint8), np.zeros(1, np.uint8), 100,
                  bm2.sw_params(bm2.default_opt(), 5))
    assert len(out) == 0


def test_bsw_long_queries_in_the_sliding_register_window(gpu_ctx_factory):
    # queries beyond 255 columns whose band fits 5 (w = 100) or 8 (w = 200) chunks of 64 columns: bsw_extend_slide (bsw_dev.h); a band of
    # 300 goes to the LDS ring.  Few pairs: the host emulator of the device sources runs this test too.
    ctx = gpu_ctx_factory()
    kw = dict(a=1, b=1, o_del=1, e_del=1, o_ins=1, e_ins=1, pen_clip5=0, pen_clip3=0)
    for seed, n, max_len, w, opts in ((51, 24, 1400, 100, kw), (52, 10, 2500, 200, kw), (53, 16, 900, 100, dict()), (54, 6, 1200, 300, kw)):
        tr = [t for t in random_pairs(seed, 4 * n, max_len=max_len, h0_max=300) if len(t[0]) > 256][:n]
        assert len(tr) >= n // 2
        _check(ctx, tr, oracle.default_opt(**opts), bm2.default_opt(**opts), w, 0 if opts else 5)


@pytest.mark.parametrize("reg_qmin", [None, "0", "64"], ids=["rows_default", "rows_in_lds", "rows_in_registers_from_64"])
def test_bsw_batch_sorted_onto_the_lane_kernel(gpu_ctx_factory, monkeypatch, reg_qmin):
    # S1 batches of >= BM2_BSW_LANES_MIN pairs are sorted on the device by (query length, target length): pairs whose query fits the lane
    # kernel's rows and whose scores fit 8-bit cells run one per LANE (the shape of getScores8 / getScores16), the others one per wavefront.
    # Same six numbers either way, in the caller's order.  (threshold lowered so that the host emulator can run it)
    monkeypatch.setenv("BM2_BSW_LANES_MIN", "64")
    if reg_qmin is not None:                                     # the classes whose rows live in registers (k_bsw_lanes<NG>, lane_dp8r): default from 96 columns
        monkeypatch.setenv("BM2_BSW_REG_QMIN", reg_qmin)
    ctx = gpu_ctx_factory()
    tr = random_pairs(61, 500, max_len=150, h0_max=100) + random_pairs(62, 60, max_len=400, h0_max=300) + random_pairs(63, 100, max_len=12, h0_max=250)
    _check(ctx, tr, oracle.default_opt(), bm2.default_opt(), 100, 5)
    monkeypatch.setenv("BM2_BSW_LANES", "0")
    _check(ctx, tr[:200], oracle.default_opt(), bm2.default_opt(), 100, 5)
