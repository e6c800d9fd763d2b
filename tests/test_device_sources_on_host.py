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
import helpers  # noqa: E402

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


def test_whole_device_pipeline_on_the_emulator(emu_lib, golden_dir):
    # every kernel of bm2_seed_chain_extend (seeding task kernels with their quad-cooperative Occ loads, SA lookup, chaining, the
    # lane-per-task extension rounds, the purge) executed by OS threads, against the oracle: 48 reads take a few seconds
    script = r'''
import sys
sys.path.insert(0, %r); sys.path.insert(0, %r); sys.path.insert(0, %r)
import numpy as np, bm2
bm2.LIB_PATH = %r
from helpers import load_golden, regs_to_records
from tools import oracle
pre, enc, off, ln, d = load_golden(%r, "g20k_l76")
n = 48
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


def test_launch_policy_knobs_do_not_change_results(emu_lib, golden_dir):
    # bm2_knob settings select kernels and code paths (wavefront-per-task extension for a query-length class, staged heavy chaining,
    # k_bwd's LDS depth / register budget, quad-cooperative SA lookup, wave-per-read purge threshold, dispatch order, round limits, the lane
    # kernel without its score table, the old stream assignment of an extension side):
    # the regs must not depend on any of them.  tools/gpu/sweep.py relies on this when it compares settings by checksum on the GPU.
    script = r'''
import sys, os
sys.path.insert(0, %r); sys.path.insert(0, %r); sys.path.insert(0, %r)
import numpy as np, bm2
bm2.LIB_PATH = %r
from helpers import load_golden, regs_to_records
from tools import oracle
pre, enc, off, ln, d = load_golden(%r, "g60k")
n = 20                                                       # (every set runs the whole device path on the emulator: seconds per read)
ln = ln[:n]; off = off[:n]; enc = enc[:int(off[-1] + ln[-1])]
ix = oracle.Index(pre); exp = ix.run(enc, off, ln)["REGPRG"].tobytes(); ix.close()
ctx = bm2.Context(0, pre)
sets = [{}, {"BM2_EXT_WAVE_QMIN": 33, "BM2_EXT_REVERSE": 1}, {"BM2_EXT_WAVE_QMIN": 161, "BM2_EXT_WAVE_NMAX": 20, "BM2_EXT_ROUNDS": 2, "BM2_EXT_PERM_SCORES": 0, "BM2_EXT_QUEUE_MAP": 0},
        {"BM2_P3_AT": 2, "BM2_BWD_EXPORT_AGE": 40},
        {"BM2_HEAVY_SA": 2, "BM2_CHAIN_STAGE": 1, "BM2_CHAIN_WAVES_PER_CU": 32, "BM2_PF_HEAVY": 2},
        {"BM2_BWD_LCAP": 4, "BM2_BWD_BLOCKS_PER_CU": 5, "BM2_BWD_WAVES": 5, "BM2_SAL_QUAD": 1},
        {"BM2_BWD_LCAP": 8, "BM2_HEAVY_SA": 5, "BM2_CHAIN_STAGE": 0, "BM2_BWD_EXPORT_AGE": 5},
        {"BM2_EXT_WAVE_QMIN": 113, "BM2_CHAIN_MAIN_SIDE": 0, "BM2_HEAVY_SA": 2, "BM2_CHAIN_CLOCK": 1},
        {"BM2_CHAIN_COOP_FLT": 1, "BM2_HEAVY_SA": 2},
        {"BM2_PERM_MODE": 4}, {"BM2_PERM_MODE": 5, "BM2_HEAVY_SA": 6}, {"BM2_PERM_MODE": 2, "BM2_HEAVY_SA": 30},      # chaining's read order: heavy first and plain / classes of seed count (default) / by histogram
        {"BM2_CHAIN_FUSE_FINISH": 0}, {"BM2_CHAIN_FUSE_FINISH": 0, "BM2_CHAIN_FINISH_PERM": 0, "BM2_CHAIN_FINISH_WAVE": 0}, {"BM2_CHAIN_FUSE_FINISH": 1, "BM2_HEAVY_SA": 4, "BM2_CHAIN_FINISH_PERM": 0},
        {"BM2_CHAIN_FINISH_WAVE": 0, "BM2_HEAVY_SA": 6}, {"BM2_CHAIN_FUSE_FINISH": 0, "BM2_HEAVY_SA": 3},   # k_chain_finish's part by k_chain's lanes (default) / by the kernel
        {"BM2_CHAIN_TIER_MAX": 1000000, "BM2_HEAVY_SA": 3}, {"BM2_CHAIN_TIER_MAX": 64, "BM2_HEAVY_SA": 3},      # every LDS tier (the routing until round 6) / the seed-rich reads to the island kernel
        {"BM2_EXT_REG_QMIN": 80}, {"BM2_EXT_REG_QMIN": 0}]       # rows in registers (lane_dp8r) for every class that has the kernel / for none (default: the 128-column class)
for kn in sets:
    for k in [k for k in os.environ if k.startswith("BM2_")]:
        del os.environ[k]
    for k, v in kn.items():
        os.environ[k] = str(v)
    regs, reg_off, st = ctx.seed_chain_extend(enc, off, ln, bm2.default_opt())
    assert regs_to_records(regs, reg_off).tobytes() == exp, kn
    if kn.get("BM2_CHAIN_CLOCK"):                              # the wavefront-per-read launches clocked their reads: counters[43..47]
        cn = ctx.batch_fetch("counters", np.uint64)
        assert len(cn) >= 48 and cn[46] > 0 and cn[47] >= cn[46], cn[40:48]       # (reads, seeds; the ticks need a real clock)
for k in [k for k in os.environ if k.startswith("BM2_")]:
    del os.environ[k]
os.environ["BM2_REF_BYTES"] = "1"                            # the reference string one base per byte on the device (read at bm2_create): the same RefPtr code, pk = 0
ctx_b = bm2.Context(0, pre)
del os.environ["BM2_REF_BYTES"]
regs, reg_off, st = ctx_b.seed_chain_extend(enc, off, ln, bm2.default_opt())
assert regs_to_records(regs, reg_off).tobytes() == exp, "BM2_REF_BYTES=1"
print("ok", len(sets))
''' % (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "bwa-mem2_amd"), emu_lib, golden_dir)
    p = subprocess.run(["python", "-c", script], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=1500)
    assert p.returncode == 0 and p.stdout.startswith(b"ok"), p.stderr.decode()[-2000:]


def test_resident_s1_batch_on_the_emulator(emu_lib):
    # bm2_bsw_upload / bm2_bsw_run / bm2_bsw_download (config 2 of bench.py: the batch stays in HBM between runs): the results of bm2_bsw,
    # the same again on a second run (the kernel writes only the output fields), the cell count of the kernel's own counter
    script = r'''
import sys
sys.path.insert(0, %r); sys.path.insert(0, %r); sys.path.insert(0, %r)
import numpy as np, bm2
bm2.LIB_PATH = %r
from helpers import pack_pairs, random_pairs
ctx = bm2.Context(0, None)
opt = bm2.default_opt()
pairs, ref, qer = pack_pairs(bm2, random_pairs(77, 60, max_len=60))
prm = bm2.sw_params(opt, 5)
exp = ctx.bsw(pairs.copy(), ref, qer, 100, prm)
ctx.bsw_upload(pairs, ref, qer)
ms, cells = ctx.bsw_run(100, prm, count_cells=True)
a = ctx.bsw_download()
ctx.bsw_run(100, prm)
b = ctx.bsw_download()
assert a.tobytes() == exp.tobytes() and b.tobytes() == exp.tobytes() and cells > 0
import bench
l2, l1, h0, q, r, qo, ro = bench.make_extension_pairs(3, 500)
assert len(q) == qo[-1] == l2.sum() and len(r) == ro[-1] == l1.sum() and (l1 >= l2).all() and h0.min() >= 19 and q.max() < 4
print("ok", cells)
''' % (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "bwa-mem2_amd"), emu_lib)
    p = subprocess.run(["python", "-c", script], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=1500)
    assert p.returncode == 0 and p.stdout.startswith(b"ok"), p.stderr.decode()[-2000:]


def test_rescue_kernel_and_sam_pe_dev_on_the_emulator(emu_lib, tmp_path):
    # k_ksw_align2 through its real launcher (task records, size-sorted order, LDS layout, list offsets) and bm2_sam_pe_dev end to end
    # (the host plans, the emulated device aligns against its reference replica, the host replays): same results as the host kernel,
    # same SAM text as the all-host path and the compiled reference.  The four row primitives are rendezvous of 16 threads here.
    script = r'''
import sys, pathlib
sys.path.insert(0, %r); sys.path.insert(0, %r); sys.path.insert(0, %r)
import numpy as np, bm2
bm2.LIB_PATH = %r
from test_ksw_align2 import _pairs, KSW_XBYTE, KSW_XSTART, KSW_XSUBO
import test_sam_tail as T
ctx = bm2.Context(0, None)
for kw in ({}, dict(a=2, b=5, o_del=7, o_ins=8, e_del=2, e_ins=1)):
    opt = bm2.default_opt(**kw)
    rng = np.random.default_rng(5)
    pairs = _pairs(41, 12) + [(rng.integers(0, 4, ql).astype(np.uint8), rng.integers(0, 4, tl).astype(np.uint8)) for ql, tl in ((1, 1), (16, 16), (17, 40), (8, 5), (33, 1))]
    xtra = [(9 * opt.a) | (KSW_XSUBO if i %% 5 else 0) | (KSW_XSTART if i %% 7 else 0) | (KSW_XBYTE if len(q) * opt.a < 250 and i %% 3 else 0) for i, (q, t) in enumerate(pairs)]
    assert (bm2.ksw_align2(pairs, xtra, opt) == bm2.ksw_align2(pairs, xtra, opt, ctx=ctx)).all()
d = pathlib.Path(%r)
fa, r1, r2 = T._pe_case(d, 61, 40, L=100, sub_rate=0.02, indel_frac=0.2, random_frac=0.05)
ref, got, pes = T._pe_run(d, fa, r1, r2, [])
stats = bm2.sam_rescue_stats()
assert ref == got and stats[0] > 20, stats
ctx2 = bm2.Context(0, fa)
ref2, got2, pes2 = T._pe_run(d, fa, r1, r2, [], ctx=ctx2)              # rescue AND CIGAR batches on the (emulated) device
assert got2 == got and bm2.sam_rescue_stats() == stats
cg = bm2.sam_cigar_stats()
assert cg[0] > 40 and cg[1] >= cg[0] and cg[2] == 0, cg      # hits in the batch, lookups it served (a hit can be printed twice), numbered hits it lacked
print("ok", stats, cg)
''' % (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "bwa-mem2_amd"), emu_lib, str(tmp_path))
    p = subprocess.run(["python", "-c", script], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=1500)
    assert p.returncode == 0 and p.stdout.startswith(b"ok"), p.stderr.decode()[-2000:]


def test_cigar_kernel_on_the_emulator(emu_lib, tmp_path):
    # k_gen_cigar (one task per lane: banded global alignment with backtrack, NM, MD) through bm2_gen_cigar_dev, against the host
    # implementation that tests/test_gen_cigar.py pins to the reference's bwa_gen_cigar2
    script = r'''
import sys, subprocess
sys.path.insert(0, %r); sys.path.insert(0, %r); sys.path.insert(0, %r)
import numpy as np, bm2
bm2.LIB_PATH = %r
from helpers import ref_binary
from tools import synth
from test_gen_cigar import make_tasks
names, ctg, alts = synth.make_genome(17, [60000, 20000], alt_contigs=0, n_repeat_families=2, repeat_len=(200, 800), copies=(3, 6),
                                     divergence=(0.0, 0.05), n_gaps=1, gap_len=(30, 100))
fa = %r
synth.write_fasta(fa, names, ctg)
subprocess.check_call([ref_binary(), "index", fa], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
ctx = bm2.Context(0, fa)
for kw in ({}, dict(a=2, b=5, o_del=7, o_ins=8, e_del=2, e_ins=1)):
    opt = bm2.default_opt(**kw)
    tasks = make_tasks(ctg, 11 + len(kw), 250)
    exp = bm2.gen_cigar(fa, opt, tasks)
    assert exp == bm2.gen_cigar(fa, opt, tasks, ctx=ctx)
    assert sum(1 for x in exp if x[2] and len(x[2]) > 1) > 100 and sum(1 for x in exp if x[2] is None) == 2
print("ok")
''' % (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "bwa-mem2_amd"), emu_lib, str(tmp_path / "g.fa"))
    from helpers import ref_binary
    if ref_binary() is None:
        helpers.no_checker("oracle/_ref reference binary not present (it builds the index)")
    p = subprocess.run(["python", "-c", script], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert p.returncode == 0 and p.stdout.startswith(b"ok"), p.stderr.decode()[-2000:]


def test_heavy_read_kernels_on_the_emulator(emu_lib, tmp_path):
    # the wave-per-item kernels that keep one repeat-rich read or start position from setting the length of a stage:
    # k_postfilter_heavy (reads with > 24 regs: the purge walk 64 regs at a time) and k_bwd_heavy (candidate lists of > 40 entries).
    # Reads from high-copy repeat families; regs / SMEMs and the backwardExt count against the oracle.
    script = r'''
import sys, os
sys.path.insert(0, %r); sys.path.insert(0, %r); sys.path.insert(0, %r)
import numpy as np, bm2
bm2.LIB_PATH = %r
from helpers import build_index, regs_to_records
from tools import oracle, refio, synth
d = %r

def case(seed, contigs, fam, n_pick, weight):
    names, ctg, alts = synth.make_genome(seed, contigs, alt_contigs=0, n_gaps=0, **fam)
    fa = os.path.join(d, "rep%%d.fa" %% seed)
    synth.write_fasta(fa, names, ctg)
    assert build_index(fa)
    reads = synth.make_reads_se(seed + 1, ctg, 400, L=150)
    enc, off, ln = refio.pack_reads(reads)
    ix = oracle.Index(fa)
    w = weight(ix.run(enc, off, ln), len(ln))
    sel = [reads[i] for i in sorted(np.argsort(-w)[:n_pick])]
    enc, off, ln = refio.pack_reads(sel)
    exp = ix.run(enc, off, ln); ix.close()
    return fa, enc, off, ln, exp

# 1. many regs per read -> k_postfilter_heavy
fa, enc, off, ln, exp = case(77, [60000, 30000], dict(n_repeat_families=2, repeat_len=(600, 900), copies=(70, 90), divergence=(0.004, 0.02)), 6,
                             lambda e, n: np.bincount(e["REGRAW"]["read"], minlength=n).astype(float))
assert np.bincount(exp["REGRAW"]["read"]).max() > 100
ctx = bm2.Context(0, fa)
regs, reg_off, st = ctx.seed_chain_extend(enc, off, ln, bm2.default_opt())
assert regs_to_records(regs, reg_off).tobytes() == exp["REGPRG"].tobytes() and st["n_ext"] == exp["counters"]["n_ext"]
# eight k_chain_heavy tiers instead of five (these reads hold hundreds of seeds: the tiers beyond the fifth) and mem_chain_flt's walk over the
# kept chains by the 64 lanes (they keep hundreds of chains)
os.environ["BM2_CHAIN_COOP_FLT"] = "1"
regs, reg_off, st = ctx.seed_chain_extend(enc, off, ln, bm2.default_opt())
del os.environ["BM2_CHAIN_COOP_FLT"]
assert regs_to_records(regs, reg_off).tobytes() == exp["REGPRG"].tobytes(), "BM2_CHAIN_COOP_FLT=1"
ctx.close()
# 2. long candidate lists -> k_bwd_heavy
def w2(e, n):
    w = np.zeros(n); np.add.at(w, e["SMEM"]["read"], e["SMEM"]["s"]); return w
fa, enc, off, ln, exp = case(91, [200000], dict(n_repeat_families=1, repeat_len=(400, 500), copies=(280, 300), divergence=(0.03, 0.05)), 6, w2)
ctx = bm2.Context(0, fa)
sm = ctx.smem(enc, off, ln, bm2.default_opt())
sc = ctx.batch_fetch("seed_counters", np.uint64)
assert int(sc[17]) + int(sc[18]) > 0, "no candidate list was long enough for k_bwd_heavy"
got = np.zeros(len(sm), refio.SMEM_DT)
for a, b in (("read", "rid"), ("m", "m"), ("n", "n"), ("k", "k"), ("l", "l"), ("s", "s")):
    got[a] = sm[b]
assert got.tobytes() == exp["SMEM"].tobytes() and int(sc[9]) == exp["counters"]["n_ext"]
print("ok")
''' % (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "bwa-mem2_amd"), emu_lib, str(tmp_path))
    p = subprocess.run(["python", "-c", script], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=1500)
    assert p.returncode == 0 and p.stdout.startswith(b"ok"), p.stderr.decode()[-2000:]

def test_long_read_chaining_kernels_on_the_emulator(emu_lib, tmp_path):
    # The long-read chaining launches (chain.hip): k_chain_islands (one wavefront per read, chaining by islands of reference buckets), the list it hands to
    # k_chain_serial (reads whose chains have EQUAL keys: chained again by the kbtree walk, internal nodes in the LDS pool -- also with a pool of two nodes, the
    # rest overflowing to global memory) and the seed filter's SW with its row in registers (k_seed_sw_reg), against the oracle.  Reads of ~0.5-1.1 kb, ONT-like
    # errors, `-x ont2d`, a repeat-rich genome; one read of the five is known (from the oracle's chains) to hold two chains with the same key.  Real long reads
    # are beyond what the thread-per-lane emulator runs in minutes: tests/test_pipeline_gpu.py has them.
    script = r"""
import sys, os
sys.path.insert(0, %r); sys.path.insert(0, %r); sys.path.insert(0, %r)
import numpy as np, bm2
bm2.LIB_PATH = %r
from helpers import build_index, regs_to_records, ONT2D
from tools import oracle, refio, synth
d = %r
names, ctg, alts = synth.make_genome(43, [400000, 150000], alt_contigs=1, alt_len=5000, n_repeat_families=40, repeat_len=(100, 2000), copies=(3, 80), divergence=(0.0, 0.03))
fa = os.path.join(d, "g.fa")
synth.write_fasta(fa, names, ctg); synth.write_alt(fa + ".alt", alts)
assert build_index(fa)
reads = synth.make_reads_long(7, ctg, 400, mean_len=1100, max_len=1400)
ix = oracle.Index(fa)
os.environ["BM2_CHAIN_TIER_MAX"] = "64"                     # no LDS tiers: every read beyond 100 seeds takes the island path
ctx = bm2.Context(0, fa)
for pick, env in (((198, 1, 3, 30, 19), {}), ((198, 1), {"BM2_CHAIN_SERIAL_LNODES": "2"})):
    sel = [reads[i] for i in pick]
    assert max(len(x) for x in sel) >= 1000                # (a chunk with a read of 1000 bases or more is a long-read chunk to the chaining stage)
    enc, off, ln = refio.pack_reads(sel)
    exp = ix.run(enc, off, ln, oracle.default_opt(**ONT2D))
    c0 = exp["CHN0"]
    dup = [r for r in range(len(ln)) if len(c0["pos"][c0["read"] == r]) != len(np.unique(c0["pos"][c0["read"] == r]))]
    assert dup == [0], dup                                  # the oracle's own chains: read 198 holds two with the same key
    for k, v in env.items(): os.environ[k] = v
    regs, reg_off, st = ctx.seed_chain_extend(enc, off, ln, bm2.default_opt(**ONT2D))
    for k in env: del os.environ[k]
    cn = ctx.batch_fetch("counters", np.uint64)
    assert regs_to_records(regs, reg_off).tobytes() == exp["REGPRG"].tobytes(), env
    assert int(cn[16]) == 1 and int(cn[39]) == 0, (int(cn[16]), int(cn[39]))      # one read listed for k_chain_serial, none of them unstaged
    if len(pick) > 2:
        assert int(cn[17]) >= 1                             # ... and at least one chained by islands
ix.close(); ctx.close()
print("ok")
""" % (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "bwa-mem2_amd"), emu_lib, str(tmp_path))
    p = subprocess.run(["python", "-c", script], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=1500)
    assert p.returncode == 0 and p.stdout.startswith(b"ok"), p.stderr.decode()[-2000:]


def test_hit_finishing_kernels_on_the_emulator(emu_lib, golden_dir, tmp_path):
    # finish.hip (mem_sort_dedup_patch + ALT flag on the device): the resumable lane-per-read walk, the wave-per-request global
    # alignment of mem_patch_reg (a fixture whose reads are split by z-drop and re-joined), the gather -- against the reference's
    # REGFIN dumps, through bm2_finish_regs_dev
    import test_finish_regs as T
    fa, enc, off, ln, d = T.split_hit_case(tmp_path)
    np.savez(str(tmp_path / "split.npz"), enc=enc, off=off, ln=ln, REGPRG=d["REGPRG"], REGFIN=d["REGFIN"])
    script = r'''
import sys
sys.path.insert(0, %r); sys.path.insert(0, %r); sys.path.insert(0, %r)
import numpy as np, bm2
bm2.LIB_PATH = %r
import test_finish_regs as T
from helpers import ONT2D, load_golden, alnregs_to_recs
def check(pre, enc, off, ln, d, kw):
    ctx = bm2.Context(0, pre)
    regs, ro = T._prg_to_regs(d["REGPRG"], len(ln))
    aln, ao = ctx.finish_regs((enc, off, ln), bm2.default_opt(**kw), regs, ro)
    ctx.close()
    assert alnregs_to_recs(aln, ao).tobytes() == d["REGFIN"].tobytes(), pre
for name, kw in T.CASES:
    pre, enc, off, ln, d = load_golden(%r, name)
    check(pre, enc, off, ln, d, kw)
z = np.load(%r)
check(%r, z["enc"], z["off"], z["ln"], z, ONT2D)
print("ok")
''' % (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "bwa-mem2_amd"), emu_lib, golden_dir, str(tmp_path / "split.npz"), fa)
    p = subprocess.run(["python", "-c", script], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=1500)
    assert p.returncode == 0 and p.stdout.startswith(b"ok"), p.stderr.decode()[-2000:]
