"""The drop-in claim, compiled: oracle/_ref/bwa-mem2.bm2 is the reference's own `bwa-mem2` (CLI, option parser, kseq reader, chunking,
header, writer) with integration/bm2_process_seqs.cpp linked in place of its mem_process_seqs (oracle/Makefile, target `bm2`).
Its SAM must equal the unmodified reference's for single-end and paired-end input, several chunks, and output-shaping options.
gpu: against the real library.  Without a GPU: the same binary with the host emulator of the device sources loaded in libbm2's
place (a check of the binding's logic: option mapping, index descriptor, text hand-over)."""
import os
import subprocess

import pytest

from helpers import ref_binary
from tools import synth
import helpers  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BM2_EXE = os.path.join(ROOT, "oracle", "_ref", "bwa-mem2.bm2")
S1_EXE = os.path.join(ROOT, "oracle", "_ref", "bwa-mem2.bm2s1")         # only seam S1 replaced (integration/bm2_bsw_binding.cpp; BASELINE config 2)


def _inputs(d, n_pairs, n_se):
    names, ctg, alts = synth.make_genome(31, [90000, 40000], alt_contigs=1, alt_len=3000, n_repeat_families=3, repeat_len=(200, 1500),
                                         copies=(3, 12), divergence=(0.0, 0.06))
    fa = os.path.join(d, "g.fa")
    synth.write_fasta(fa, names, ctg)
    synth.write_alt(fa + ".alt", alts)
    subprocess.check_call([ref_binary(), "index", fa], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    r1, r2 = synth.make_reads_pe(32, ctg, n_pairs, L=120)
    synth.write_fastq(os.path.join(d, "r1.fq"), r1, suffix="/1")
    synth.write_fastq(os.path.join(d, "r2.fq"), r2, suffix="/2")
    synth.write_fastq(os.path.join(d, "se.fq"), synth.make_reads_se(33, ctg, n_se, L=100))
    return fa


def _compare(d, fa, env, K, exe=BM2_EXE, marker=b"libbm2", cases=("pe", "se", "pe_opts")):
    pe = [os.path.join(d, "r1.fq"), os.path.join(d, "r2.fq")]
    for tag, args in (("pe", ["-K", str(K), fa] + pe), ("se", [fa, os.path.join(d, "se.fq")]),
                      ("pe_opts", ["-R", "@RG\\tID:x\\tSM:y", "-Y", "-M", "-a", fa] + pe)):
        if tag not in cases:
            continue
        a = subprocess.run([ref_binary(), "mem", "-t", "4"] + args, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, check=True).stdout
        p = subprocess.run([exe, "mem", "-t", "4"] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
        assert p.returncode == 0, p.stderr.decode()[-1500:]
        la = [l for l in a.split(b"\n") if not l.startswith(b"@PG")]
        lb = [l for l in p.stdout.split(b"\n") if not l.startswith(b"@PG")]
        assert len(la) > 50
        for i, (x, y) in enumerate(zip(la, lb)):
            assert x == y, "%s line %d\n  ref : %s\n  bm2 : %s" % (tag, i, x[:300], y[:300])
        assert len(la) == len(lb), tag
        assert marker in p.stderr               # the chunk really went through the replacement


def _need(exe=BM2_EXE):
    if ref_binary() is None or not os.path.exists(exe):
        helpers.no_checker("oracle/_ref (reference + %s) not built: make -C oracle ref bm2 bm2s1" % os.path.basename(exe))


def test_binding_against_the_emulator(tmp_path, emu_lib):
    _need()
    from conftest import emu_dir_as_libbm2
    d = emu_dir_as_libbm2(emu_lib, tmp_path)                        # LD_LIBRARY_PATH goes before the binary's RUNPATH
    fa = _inputs(str(tmp_path), 150, 100)
    _compare(str(tmp_path), fa, dict(os.environ, LD_LIBRARY_PATH=d), 20000)


@pytest.mark.gpu
def test_binding_on_the_gpu(tmp_path):
    _need()
    fa = _inputs(str(tmp_path), 3000, 2000)
    _compare(str(tmp_path), fa, dict(os.environ), 300000)


def test_s1_binding_against_the_emulator(tmp_path, emu_lib):
    """bwa-mem2.bm2s1: the reference with ONLY its banded extension (getScores8 / getScores16 / scalarBandedSWAWrapper) routed to bm2_bsw;
    seeding, chaining, the band retry rule, pairing and SAM are the reference's host code."""
    _need(S1_EXE)
    from conftest import emu_dir_as_libbm2
    d = emu_dir_as_libbm2(emu_lib, tmp_path)
    fa = _inputs(str(tmp_path), 60, 52)                             # (every extension batch of the reference's 512-read blocks goes through the emulator: minutes per 100 pairs)
    _compare(str(tmp_path), fa, dict(os.environ, LD_LIBRARY_PATH=d, BM2_S1_CONTEXTS="2"), 9000, S1_EXE, b"[bm2s1]", cases=("pe", "se"))   # (the option set of the third case does not reach this seam; the GPU test runs all three)


@pytest.mark.gpu
def test_s1_binding_on_the_gpu(tmp_path):
    _need(S1_EXE)
    fa = _inputs(str(tmp_path), 3000, 2000)
    _compare(str(tmp_path), fa, dict(os.environ), 300000, S1_EXE, b"[bm2s1]")
