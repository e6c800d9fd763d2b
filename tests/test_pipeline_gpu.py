"""Hot-path parity on the GPU: every stage of libbm2's device pipeline against the reference dumps (golden fixtures)
and against the oracle on freshly generated inputs.  Bit-exact; calls go through the C ABI (bm2.py is a thin ctypes
binding)."""
import os

import numpy as np
import pytest

import bm2
from helpers import (ONT2D, build_index, chain_mask, first_diff, gpu_stage_records, load_golden, regs_to_records)
from tools import oracle, refio, synth
import helpers  # noqa: E402

pytestmark = pytest.mark.gpu


def _same(exp, got, what):
    assert len(exp) == len(got) and exp.tobytes() == got.tobytes(), "%s: %s" % (what, first_diff(exp, got))


@pytest.mark.parametrize("name", ["g60k", "g20k_l76"])
def test_golden_all_stages(gpu_ctx_factory, golden_dir, name):
    pre, enc, off, ln, d = load_golden(golden_dir, name)
    ctx = gpu_ctx_factory(pre)
    opt = bm2.default_opt()
    # S2: SMEMs, sorted (rid, m, n)
    sm = ctx.smem(enc, off, ln, opt)
    exp = d["SMEM"]
    got = np.zeros(len(sm), refio.SMEM_DT)
    for a, b in (("read", "rid"), ("m", "m"), ("n", "n"), ("k", "k"), ("l", "l"), ("s", "s")):
        got[a] = sm[b]
    _same(exp, got, "SMEM")
    # S2: SA coordinates in (SMEM, occurrence) order
    co = ctx.sal(sm, opt.max_occ)
    _same(d["SACOORD"], co, "SACOORD")
    # S3
    regs, reg_off, st = ctx.seed_chain_extend(enc, off, ln, opt)
    C, S, R = gpu_stage_records(ctx, bm2, len(ln))
    _same(d["CHN1"], C, "CHN1")
    _same(d["SEED1"], S, "SEED1")
    # regs before the purge: the device decides redundancy BEFORE extending (lazy rounds), the reference after; purged
    # regs therefore agree on the purge marker only, surviving regs on every field
    exp_raw = d["REGRAW"]
    assert len(exp_raw) == len(R)
    purged = (exp_raw["qb"] == -1) & (exp_raw["qe"] == -1)
    assert ((R["qb"] == -1) & (R["qe"] == -1) == purged).all()
    _same(exp_raw[~purged], R[~purged], "REGRAW (kept)")
    _same(d["REGPRG"], regs_to_records(regs, reg_off), "REGPRG")
    assert st["n_smem"] == len(d["SMEM"]) and st["n_sa"] == len(d["SACOORD"]) and st["n_reg"] == len(d["REGPRG"])


def test_empty_and_degenerate_batches(gpu_ctx_factory, golden_dir):
    pre, enc, off, ln, d = load_golden(golden_dir, "g20k_l76")
    ctx = gpu_ctx_factory(pre)
    opt = bm2.default_opt()
    regs, reg_off, st = ctx.seed_chain_extend(np.zeros(0, np.uint8), np.zeros(0, np.int64), np.zeros(0, np.int32), opt)
    assert len(regs) == 0 and list(reg_off) == [0]
    # reads that cannot seed: all N, shorter than min_seed_len, length 1
    seqs = [np.full(50, 4, np.uint8), np.array([0, 1, 2, 3, 0, 1], np.uint8), np.array([2], np.uint8)]
    e, o, l = refio.pack_reads(seqs)
    regs, reg_off, st = ctx.seed_chain_extend(e, o, l, opt)
    assert len(regs) == 0 and list(reg_off) == [0, 0, 0, 0]
    assert len(ctx.smem(e, o, l, opt)) == 0


def _fresh_case(tmp_path, seed, contigs, n_reads, L, **kw):
    names, ctg, alts = synth.make_genome(seed, contigs, alt_contigs=1, alt_len=3000, n_repeat_families=6,
                                         repeat_len=(200, 3000), copies=(3, 40), divergence=(0.0, 0.08))
    fa = str(tmp_path / "g.fa")
    synth.write_fasta(fa, names, ctg)
    if alts:
        synth.write_alt(fa + ".alt", alts)
    if not build_index(fa):
        helpers.no_checker("oracle/_ref reference binary not present (build it with `make -C oracle ref`)")
    reads = synth.make_reads_se(seed + 1, ctg, n_reads, L=L, **kw)
    return fa, refio.pack_reads(reads)


@pytest.mark.parametrize("seed,L,n", [(5, 150, 6000), (6, 101, 3000), (7, 250, 2000)])
def test_fresh_inputs_vs_oracle(gpu_ctx_factory, tmp_path, seed, L, n):
    fa, (enc, off, ln) = _fresh_case(tmp_path, seed, [300000, 150000, 50000], n, L)
    ix = oracle.Index(fa)
    try:
        exp = ix.run(enc, off, ln)
    finally:
        ix.close()
    ctx = gpu_ctx_factory(fa)
    regs, reg_off, st = ctx.seed_chain_extend(enc, off, ln, bm2.default_opt())
    _same(exp["REGPRG"], regs_to_records(regs, reg_off), "REGPRG")
    c = exp["counters"]
    assert st["n_ext"] == c["n_ext"] and st["n_lf"] == c["n_lf"]
    # lazy rounds skip the extension of seeds the reference extends and then purges
    assert 0 < st["n_sw_cells"] <= c["n_sw_cells"] and 0 < st["n_sw_tasks"] <= len(exp["PAIR"])


def test_non_default_options_vs_oracle(gpu_ctx_factory, tmp_path):
    fa, (enc, off, ln) = _fresh_case(tmp_path, 9, [120000, 60000], 2500, 150, sub_rate=0.03, indel_frac=0.3)
    kw = dict(min_seed_len=15, w=30, max_occ=50, zdrop=150, b=3, o_del=4, o_ins=5, e_del=2, e_ins=1, pen_clip5=3, pen_clip3=7,
              max_mem_intv=0, split_width=5, drop_ratio=0.3, max_chain_gap=500)
    ix = oracle.Index(fa)
    try:
        exp = ix.run(enc, off, ln, oracle.default_opt(**kw))
    finally:
        ix.close()
    ctx = gpu_ctx_factory(fa)
    regs, reg_off, st = ctx.seed_chain_extend(enc, off, ln, bm2.default_opt(**kw))
    _same(exp["REGPRG"], regs_to_records(regs, reg_off), "REGPRG")


def _repeat_case(tmp_path, seed, n_reads):
    # a few high-copy, low-divergence repeat families: candidate lists of the backward phases run to > 100 entries
    names, ctg, alts = synth.make_genome(seed, [400000, 200000], alt_contigs=0, n_repeat_families=4, repeat_len=(1500, 4000),
                                         copies=(150, 300), divergence=(0.005, 0.04))
    fa = str(tmp_path / "rep.fa")
    synth.write_fasta(fa, names, ctg)
    if not build_index(fa):
        helpers.no_checker("oracle/_ref reference binary not present (build it with `make -C oracle ref`)")
    return fa, refio.pack_reads(synth.make_reads_se(seed + 1, ctg, n_reads, L=150))


def test_repeat_rich_lists_vs_oracle(gpu_ctx_factory, tmp_path):
    # exercises what ordinary reads rarely do: lists beyond the 32-entry slot (pool), beyond the 12 LDS survivors, and the
    # wave-per-task kernel for lists > 40
    fa, (enc, off, ln) = _repeat_case(tmp_path, 21, 5000)
    ix = oracle.Index(fa)
    try:
        exp = ix.run(enc, off, ln)
    finally:
        ix.close()
    ctx = gpu_ctx_factory(fa)
    regs, reg_off, st = ctx.seed_chain_extend(enc, off, ln, bm2.default_opt())
    sc = ctx.batch_fetch("seed_counters", np.uint64)
    assert sc[11] > 0 and sc[17] > 0 and sc[18] > 0, "the case must reach the pool and the heavy-task kernel (%s)" % sc
    _same(exp["REGPRG"], regs_to_records(regs, reg_off), "REGPRG")
    assert st["n_ext"] == exp["counters"]["n_ext"]
    sm = ctx.smem(enc, off, ln, bm2.default_opt())
    got = np.zeros(len(sm), refio.SMEM_DT)
    for a, b in (("read", "rid"), ("m", "m"), ("n", "n"), ("k", "k"), ("l", "l"), ("s", "s")):
        got[a] = sm[b]
    _same(exp["SMEM"], got, "SMEM")


# Launch-policy knobs that are OFF by default (bm2_knob, bm2_ctx.h): none may change a result, and none may sit in the tree without having met a GPU.
# Every setting below runs the repeat-rich short reads (pool lists, wavefront-per-task seeding, the heavy chaining tiers, wave / lane extension
# classes) and a set of long ONT-like reads (island chaining, serial equal-key reads, the sliding-window extension) against the oracle.
KNOB_SETTINGS = [
    {"BM2_BWD_EXPORT_AGE": "256"}, {"BM2_BWD_EXPORT_AGE": "24"}, {"BM2_BWD_EXPORT_AGE": "128"},
    {"BM2_BWD_EXPORT_AGE": "0"}, {"BM2_BWD_EXPORT_AGE": "64", "BM2_BWD_LCAP": "8", "BM2_BWD_BLOCKS_PER_CU": "4"}, {"BM2_BWD_CONT_BPC": "2", "BM2_BWD_EXPORT_AGE": "100"},
    {"BM2_P3_BPC": "1"}, {"BM2_P3_BPC": "2", "BM2_P3_AT": "2"},
    {"BM2_CHAIN_COOP_FLT": "0"}, {"BM2_CHAIN_COOP_FLT": "1"}, {"BM2_CHAIN_CLOCK": "1"}, {"BM2_CHAIN_HEAVY_WPE": "2"}, {"BM2_CHAIN_HEAVY_WPE": "4", "BM2_CHAIN_COOP_FLT": "0"},
    {"BM2_CHAIN_SERIAL_LNODES": "3", "BM2_CHAIN_ISL_WPE": "3"}, {"BM2_CHAIN_SERIAL_LNODES": "400", "BM2_CHAIN_COOP_FLT": "1"},
    {"BM2_EXT_REG_QMIN": "80"}, {"BM2_EXT_REG_QMIN": "0"}, {"BM2_KSW_REG": "0"},
    {"BM2_PERM_MODE": "4"}, {"BM2_PERM_MODE": "2"}, {"BM2_PERM_MODE": "5", "BM2_HEAVY_SA": "64"},
    {"BM2_CHAIN_FUSE_FINISH": "0"}, {"BM2_CHAIN_FUSE_FINISH": "0", "BM2_CHAIN_FINISH_PERM": "0", "BM2_CHAIN_FINISH_WAVE": "0"}, {"BM2_CHAIN_FINISH_PERM": "0", "BM2_HEAVY_SA": "8"},
    {"BM2_CHAIN_FINISH_WAVE": "0"}, {"BM2_CHAIN_TIER_MAX": "1000000"}, {"BM2_CHAIN_TIER_MAX": "128", "BM2_HEAVY_SA": "100"},
]


@pytest.fixture(scope="module")
def knob_cases(tmp_path_factory):
    tmp = tmp_path_factory.mktemp("knobs")
    fa, (enc, off, ln) = _repeat_case(tmp, 23, 4000)
    ix = oracle.Index(fa)
    try:
        exp_short = ix.run(enc, off, ln)
    finally:
        ix.close()
    (tmp / "long").mkdir()
    names, ctg, alts = synth.make_genome(47, [400000, 150000], alt_contigs=1, alt_len=5000, n_repeat_families=40, repeat_len=(100, 2000),
                                         copies=(3, 80), divergence=(0.0, 0.03))
    fl = str(tmp / "long" / "g.fa")
    synth.write_fasta(fl, names, ctg)
    synth.write_alt(fl + ".alt", alts)
    assert build_index(fl)
    lenc, loff, lln = refio.pack_reads(synth.make_reads_long(48, ctg, 48, mean_len=5000, max_len=20000))
    ix = oracle.Index(fl)
    try:
        exp_long = ix.run(lenc, loff, lln, oracle.default_opt(**ONT2D))
    finally:
        ix.close()
    return (fa, enc, off, ln, exp_short), (fl, lenc, loff, lln, exp_long)


@pytest.mark.parametrize("env", KNOB_SETTINGS, ids=lambda e: ",".join("%s=%s" % kv for kv in e.items()))
def test_off_by_default_knobs_keep_every_result(gpu_ctx_factory, knob_cases, monkeypatch, env):
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    (fa, enc, off, ln, exp), (fl, lenc, loff, lln, lexp) = knob_cases
    ctx = gpu_ctx_factory(fa)
    for rep in range(2):                                     # (twice: the second batch runs with the first one's launch statistics)
        regs, reg_off, st = ctx.seed_chain_extend(enc, off, ln, bm2.default_opt())
        _same(exp["REGPRG"], regs_to_records(regs, reg_off), "REGPRG (short reads, batch %d)" % rep)
        assert st["n_ext"] == exp["counters"]["n_ext"]
    sc = ctx.batch_fetch("seed_counters", np.uint64)
    if env.get("BM2_BWD_EXPORT_AGE", "256") != "0":
        assert int(sc[21]) > 0 and int(sc[22]) > 0, "no backward task was handed over (%s)" % sc
    ctx = gpu_ctx_factory(fl)
    regs, reg_off, st = ctx.seed_chain_extend(lenc, loff, lln, bm2.default_opt(**ONT2D))
    _same(lexp["REGPRG"], regs_to_records(regs, reg_off), "REGPRG (long reads)")
    assert st["n_ext"] == lexp["counters"]["n_ext"]


def test_seeding_workspace_growth(gpu_ctx_factory, tmp_path, monkeypatch):
    # start from workspaces that hold nothing but the pool tails: the run must report what it needs, grow, and repeat
    fa, (enc, off, ln) = _fresh_case(tmp_path, 31, [200000, 80000], 40000, 150)
    ctx = gpu_ctx_factory(fa)
    r0, o0, _ = ctx.seed_chain_extend(enc, off, ln, bm2.default_opt())
    assert ctx.batch_fetch("seed_attempts", np.int32)[0] == 1
    monkeypatch.setenv("BM2_SEED_TINY", "1")
    ctx2 = gpu_ctx_factory(fa)
    r1, o1, _ = ctx2.seed_chain_extend(enc, off, ln, bm2.default_opt())
    assert ctx2.batch_fetch("seed_attempts", np.int32)[0] > 1
    assert r0.tobytes() == r1.tobytes() and o0.tobytes() == o1.tobytes()


def test_split_api_is_idempotent(gpu_ctx_factory, golden_dir):
    # upload once, run twice: identical regs (no state leaks between runs of a resident batch)
    pre, enc, off, ln, d = load_golden(golden_dir, "g60k")
    ctx = gpu_ctx_factory(pre)
    opt = bm2.default_opt()
    ctx.batch_upload(enc, off, ln)
    ctx.batch_run(opt)
    r1, o1 = ctx.batch_download()
    ctx.batch_run(opt)
    r2, o2 = ctx.batch_download()
    assert r1.tobytes() == r2.tobytes() and o1.tobytes() == o2.tobytes()
    stages = []
    for n, _ in ctx.batch_kernel_ms():                  # timed intervals are "stage" or "stage.kernel"
        if n.split(".")[0] not in stages:
            stages.append(n.split(".")[0])
    assert stages == ["smem", "sal", "chain", "extend", "postfilter"]


def test_sub_batch_pipelining_matches_single_part(tmp_path):
    # a chunk big enough to be cut into 4 parts (at multiples of 512 reads), each driven by its own host thread
    fa, (enc, off, ln) = _fresh_case(tmp_path, 13, [200000, 100000], 140000, 100)
    os.environ["BM2_N_SUB"] = "1"
    try:
        c1 = bm2.Context(0, fa)
    finally:
        del os.environ["BM2_N_SUB"]; os.environ.pop("BM2_SUB_STAGGER", None)
    try:
        r1, o1, s1 = c1.seed_chain_extend(enc, off, ln, bm2.default_opt())
    finally:
        c1.close()
    os.environ["BM2_N_SUB"] = "4"; os.environ["BM2_SUB_STAGGER"] = "1"      # staggered by the stage gate: part i + 1 seeds while part i extends
    try:
        c4 = bm2.Context(0, fa)
        try:
            r4, o4, s4 = c4.seed_chain_extend(enc, off, ln, bm2.default_opt())
            kms = c4.batch_kernel_ms()
        finally:
            c4.close()
    finally:
        del os.environ["BM2_N_SUB"]; os.environ.pop("BM2_SUB_STAGGER", None)
    assert o1.tobytes() == o4.tobytes() and r1.tobytes() == r4.tobytes()
    assert s1 == s4 and len({n.split(".")[0] for n, _ in kms}) == 5
    ix = oracle.Index(fa)
    try:
        exp = ix.run(enc, off, ln)
    finally:
        ix.close()
    _same(exp["REGPRG"], regs_to_records(r4, o4), "REGPRG")


@pytest.mark.parametrize("name,L", [("g60k", 150), ("g20k_l76", 76)])
def test_extension_lane_kernel_without_the_score_table(golden_dir, name, L, tmp_path):
    # the lane kernels take a cell's score from a byte permute over the row's score table (the default when the three scores fit a
    # signed byte); BM2_EXT_PERM_SCORES=0 is the kernel they fall back to -- 4-bit query, two compares per cell: the goldens' regs
    # through it, and a fresh chunk of reads whose query-length classes fill whole wavefronts through both
    pre, enc, off, ln, d = load_golden(golden_dir, name)
    os.environ["BM2_EXT_PERM_SCORES"] = "0"
    try:
        ctx = bm2.Context(0, pre)
        try:
            regs, reg_off, st = ctx.seed_chain_extend(enc, off, ln, bm2.default_opt())
        finally:
            ctx.close()
        _same(d["REGPRG"], regs_to_records(regs, reg_off), "REGPRG")
        fa, (enc2, off2, ln2) = _fresh_case(tmp_path, 31, [200000, 100000], 3000, L)
        c1 = bm2.Context(0, fa)
        try:
            r1, o1, s1 = c1.seed_chain_extend(enc2, off2, ln2, bm2.default_opt())
        finally:
            c1.close()
    finally:
        del os.environ["BM2_EXT_PERM_SCORES"]
    c0 = bm2.Context(0, fa)
    try:
        r0, o0, s0 = c0.seed_chain_extend(enc2, off2, ln2, bm2.default_opt())
    finally:
        c0.close()
    assert o0.tobytes() == o1.tobytes() and r0.tobytes() == r1.tobytes() and s0 == s1


def test_long_reads_ont2d_golden(gpu_ctx_factory, golden_dir):
    # config-5 shape: the seed filter (local SW per short seed), chains emptied by it, kb-long int16/int32-class extensions
    pre, enc, off, ln, d = load_golden(golden_dir, "g40k_ont")
    ctx = gpu_ctx_factory(pre)
    opt = bm2.default_opt(**ONT2D)
    regs, reg_off, st = ctx.seed_chain_extend(enc, off, ln, opt)
    C, S, R = gpu_stage_records(ctx, bm2, len(ln))
    _same(d["CHN1"], C, "CHN1")
    _same(d["SEED1"], S, "SEED1")
    _same(d["REGPRG"], regs_to_records(regs, reg_off), "REGPRG")


def test_long_reads_smem_order_in_runs_of_start_classes(golden_dir):
    # k_smem_finish_big with its LDS capacity cut to 128 keys: the SMEM-rich reads of the fixture are ordered in runs of m-classes (the path a 30 kb
    # read takes at the real capacity), a class that alone exceeds the capacity falls back to the one-lane sort
    pre, enc, off, ln, d = load_golden(golden_dir, "g40k_ont")
    os.environ["BM2_SMEM_SORT_KEYS"] = "128"
    try:
        ctx = bm2.Context(0, pre)
        try:
            regs, reg_off, st = ctx.seed_chain_extend(enc, off, ln, bm2.default_opt(**ONT2D))
            sm = ctx.smem(enc, off, ln, bm2.default_opt(**ONT2D))
        finally:
            ctx.close()
    finally:
        del os.environ["BM2_SMEM_SORT_KEYS"]
    got = np.zeros(len(sm), refio.SMEM_DT)
    for a, b in (("read", "rid"), ("m", "m"), ("n", "n"), ("k", "k"), ("l", "l"), ("s", "s")):
        got[a] = sm[b]
    _same(d["SMEM"], got, "SMEM")
    _same(d["REGPRG"], regs_to_records(regs, reg_off), "REGPRG")


def test_long_reads_fresh_vs_oracle(gpu_ctx_factory, tmp_path):
    names, ctg, alts = synth.make_genome(41, [200000, 90000], alt_contigs=1, alt_len=4000, n_repeat_families=5,
                                         repeat_len=(300, 4000), copies=(3, 20), divergence=(0.0, 0.08))
    fa = str(tmp_path / "g.fa")
    synth.write_fasta(fa, names, ctg)
    synth.write_alt(fa + ".alt", alts)
    if not build_index(fa):
        helpers.no_checker("oracle/_ref reference binary not present")
    reads = synth.make_reads_long(42, ctg, 60, mean_len=3000, max_len=9000)
    enc, off, ln = refio.pack_reads(reads)
    ix = oracle.Index(fa)
    try:
        exp = ix.run(enc, off, ln, oracle.default_opt(**ONT2D))
    finally:
        ix.close()
    ctx = gpu_ctx_factory(fa)
    regs, reg_off, st = ctx.seed_chain_extend(enc, off, ln, bm2.default_opt(**ONT2D))
    _same(exp["REGPRG"], regs_to_records(regs, reg_off), "REGPRG")


def test_long_reads_at_scale_against_the_reference(gpu_ctx_factory, tmp_path):
    # config-5 shape at its real sizes: 200 reads of mean 10 kb, one of them at the 30 kb cap (> 16 k SMEMs: the workgroup sort of k_smem_finish_big at
    # its real LDS capacity, runs of start classes; > 30 k seeds: the island kernel and the sliding register window on their longest inputs), against
    # the dump of the COMPILED REFERENCE (oracle/_ref/refdump: the reference's own mem_kernel1_core / mem_kernel2_core) -- the oracle's single
    # thread would need minutes for these reads
    import subprocess
    from helpers import ref_binary
    refdump = ref_binary("refdump")
    if refdump is None:
        helpers.no_checker("oracle/_ref not built")
    names, ctg, alts = synth.make_genome(45, [1800000, 900000, 300000], alt_contigs=1, alt_len=20000, n_repeat_families=30, repeat_len=(300, 5000),
                                         copies=(5, 60), divergence=(0.01, 0.12))
    fa = str(tmp_path / "g.fa")
    synth.write_fasta(fa, names, ctg)
    synth.write_alt(fa + ".alt", alts)
    if not build_index(fa):
        helpers.no_checker("oracle/_ref reference binary not present")
    reads = synth.make_reads_long(46, ctg, 200, mean_len=10000, max_len=30000)
    assert max(len(r) for r in reads) >= 25000 and np.mean([len(r) for r in reads]) > 8000
    enc, off, ln = refio.pack_reads(reads)
    rtxt, dump = str(tmp_path / "reads.txt"), str(tmp_path / "dump.bin")
    acgtn = np.frombuffer(b"ACGTN", np.uint8)
    with open(rtxt, "wb") as f:
        for r in reads:
            f.write(acgtn[r].tobytes() + b"\n")
    p = subprocess.run([refdump, "-x", "ont2d", fa, rtxt, dump], stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True)
    assert p.returncode == 0, p.stderr[-400:]
    exp = refio.read_dump(dump)
    ctx = gpu_ctx_factory(fa)
    regs, reg_off, st = ctx.seed_chain_extend(enc, off, ln, bm2.default_opt(**ONT2D))
    _same(exp["REGPRG"], regs_to_records(regs, reg_off), "REGPRG")
    assert st["n_sa"] / len(reads) > 1000          # the reads are seed-rich enough for the long-read kernels


@pytest.mark.parametrize("env", [{}, {"BM2_CHAIN_SERIAL_LNODES": "2"}, {"BM2_CHAIN_COOP_FLT": "1", "BM2_CHAIN_ISL_WPE": "3"}, {"BM2_CHAIN_SERIAL_BESIDE": "0", "BM2_CHAIN_SERIAL_HYB": "0"}],
                         ids=lambda e: ",".join("%s=%s" % kv for kv in e.items()) or "default")
def test_long_reads_chained_by_islands(gpu_ctx_factory, tmp_path, monkeypatch, env):
    # mem_chain_seeds of seed-rich reads cut into islands of reference buckets (k_chain_islands, chain.hip): a repeat-rich genome, so that a read
    # brings hundreds of stray hits -- islands of one seed -- beside its locus; the tiers of the wavefront-per-read kernel are switched off so
    # that every read beyond 100 seeds takes the island path.  Chains, seeds and regs must equal the oracle's; the kernel says how many reads it
    # chained by islands and how many it handed to the serial code (equal chain keys).
    # (the reads with equal chain keys: k_chain_serial -- the tree's internal nodes in LDS -- by default; inside the island kernel; with an LDS pool of two nodes)
    monkeypatch.setenv("BM2_CHAIN_TIER_MAX", "64")
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    names, ctg, alts = synth.make_genome(43, [400000, 150000], alt_contigs=1, alt_len=5000, n_repeat_families=40, repeat_len=(100, 2000),
                                         copies=(3, 80), divergence=(0.0, 0.03))
    fa = str(tmp_path / "g.fa")
    synth.write_fasta(fa, names, ctg)
    synth.write_alt(fa + ".alt", alts)
    if not build_index(fa):
        helpers.no_checker("oracle/_ref reference binary not present")
    reads = synth.make_reads_long(44, ctg, 40, mean_len=5000, max_len=20000)
    enc, off, ln = refio.pack_reads(reads)
    ix = oracle.Index(fa)
    try:
        exp = ix.run(enc, off, ln, oracle.default_opt(**ONT2D))
    finally:
        ix.close()
    ctx = gpu_ctx_factory(fa)
    regs, reg_off, st = ctx.seed_chain_extend(enc, off, ln, bm2.default_opt(**ONT2D))
    cn = ctx.batch_fetch("counters", np.uint64)
    C, S, R = gpu_stage_records(ctx, bm2, len(ln))
    _same(chain_mask(exp["CHN1"]), chain_mask(C), "CHN1")
    _same(exp["SEED1"], S, "SEED1")
    _same(exp["REGPRG"], regs_to_records(regs, reg_off), "REGPRG")
    assert int(cn[17]) + int(cn[16]) >= 30 and int(cn[18]) >= int(cn[17]), (int(cn[16]), int(cn[17]), int(cn[18]))
    print("island path: %d reads, %d islands, %d reads chained serially (equal keys)" % (int(cn[17]), int(cn[18]), int(cn[16])))


@pytest.mark.parametrize("kw", [dict(e_del=2, e_ins=3), dict(zdrop=0), dict(zdrop=200),
                                dict(a=2, b=8, o_del=12, o_ins=12, e_del=2, e_ins=2, zdrop=200, pen_clip5=10, pen_clip3=10)])
def test_zdrop_rule_of_the_simd_kernels(gpu_ctx_factory, tmp_path, kw):
    # option sets where the Z-drop test of the reference's int8 / int16 kernels differs from the scalar one (the last = `-A 2`)
    fa, (enc, off, ln) = _fresh_case(tmp_path, 19, [100000, 40000], 2000, 150, sub_rate=0.03, indel_frac=0.3)
    ix = oracle.Index(fa)
    try:
        exp = ix.run(enc, off, ln, oracle.default_opt(**kw))
    finally:
        ix.close()
    ctx = gpu_ctx_factory(fa)
    regs, reg_off, st = ctx.seed_chain_extend(enc, off, ln, bm2.default_opt(**kw))
    _same(exp["REGPRG"], regs_to_records(regs, reg_off), "REGPRG")


def test_non_bwa_scoring_matrix_is_refused(gpu_ctx_factory, golden_dir):
    pre, enc, off, ln, d = load_golden(golden_dir, "g20k_l76")
    ctx = gpu_ctx_factory(pre)
    opt = bm2.default_opt()
    opt.mat[7] = -2              # C vs G scored differently from the other mismatches
    with pytest.raises(bm2.Bm2Error):
        ctx.seed_chain_extend(enc, off, ln, opt)
