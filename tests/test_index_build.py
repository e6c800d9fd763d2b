"""bm2_index_build (multi-threaded host index builder) writes the same bytes as `bwa-mem2 index`.  No GPU."""
import filecmp
import os
import subprocess

import numpy as np
import pytest

import bm2
from helpers import ref_binary
from tools import synth
import helpers  # noqa: E402


def test_index_matches_reference_index_bytes(tmp_path):
    exe = ref_binary()
    if exe is None:
        helpers.no_checker("oracle/_ref not built")
    names, ctg, alts = synth.make_genome(17, [70000, 30001, 999], n_repeat_families=3, repeat_len=(200, 2000), copies=(3, 9),
                                         divergence=(0.0, 0.05), n_gaps=3, gap_len=(1, 300), alt_contigs=1, alt_len=2000)
    ref_fa, my_fa = str(tmp_path / "ref.fa"), str(tmp_path / "mine.fa")
    synth.write_fasta(ref_fa, names, ctg)
    synth.write_fasta(my_fa, names, ctg)
    subprocess.check_call([exe, "index", ref_fa], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    bm2.index_build(my_fa, None, 3)
    for ext in (".pac", ".ann", ".amb", ".0123", ".bwt.2bit.64"):
        assert filecmp.cmp(ref_fa + ext, my_fa + ext, shallow=False), ext


def test_index_rebuilds_golden_fixture(golden_dir, tmp_path):
    # the fixture's .0123 has its N already replaced by bases: rebuilding from it must give the fixture's FM-index bytes
    pre = os.path.join(golden_dir, "g20k_l76.fa")
    t = np.fromfile(pre + ".0123", dtype=np.uint8)
    fwd = t[:len(t) // 2]
    lens = [int(l.split()[1]) for i, l in enumerate(open(pre + ".ann").read().split("\n")[1:]) if i % 2 == 1 and l]
    fa = str(tmp_path / "g.fa")
    ctg, o = [], 0
    for l in lens:
        ctg.append(fwd[o:o + l]); o += l
    synth.write_fasta(fa, ["chr%d" % (i + 1) for i in range(len(ctg))], ctg)
    bm2.index_build(fa, None, 2)
    for ext in (".pac", ".0123", ".bwt.2bit.64"):
        assert filecmp.cmp(pre + ext, fa + ext, shallow=False), ext


def test_index_build_errors(tmp_path):
    with pytest.raises(bm2.Bm2Error):
        bm2.index_build(str(tmp_path / "missing.fa"))
    p = tmp_path / "empty.fa"
    p.write_text(">x\n")
    with pytest.raises(bm2.Bm2Error):
        bm2.index_build(str(p))


def test_index_odd_fasta_text_and_long_alt_lines(tmp_path):
    # kseq keeps every character of a sequence line except the CR of a CR-LF end: a blank or a tab inside a line is an ambiguous base
    # (random fill + a hole in .amb); and the .alt reader takes the first field of each line, however long the rest (hs38DH.fa.alt
    # holds whole SAM records)
    exe = ref_binary()
    if exe is None:
        helpers.no_checker("oracle/_ref not built")
    rng = np.random.default_rng(4)
    def seq(n):
        return "".join("ACGT"[i] for i in rng.integers(0, 4, n))
    text = ">c1 first contig\r\n" + seq(70) + "\r\n" + seq(30) + " " + seq(39) + "\r\n" + seq(10) + "\t" + seq(9) + "NNNN" + seq(40) + "\n" + \
           ">c2\n" + seq(500) + "\n" + seq(123) + "\n>c2_alt\n" + seq(300) + "\n"
    alt = "@HD\tVN:1.0\n" + "c2_alt\t0\tc2\t1\t60\t" + "10M" * 4000 + "\t*\t0\t0\t" + "A" * 30000 + "\n" + "nosuch\t0\n"
    ref_fa, my_fa = str(tmp_path / "ref.fa"), str(tmp_path / "mine.fa")
    for p in (ref_fa, my_fa):
        open(p, "w", newline="").write(text)
        open(p + ".alt", "w").write(alt)
    subprocess.check_call([exe, "index", ref_fa], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    bm2.index_build(my_fa, None, 2)
    for ext in (".pac", ".ann", ".amb", ".0123", ".bwt.2bit.64"):
        assert filecmp.cmp(ref_fa + ext, my_fa + ext, shallow=False), ext
    d = bm2.Index(my_fa)
    try:
        import ctypes as C
        alts = [C.cast(d._desc.ann_is_alt, C.POINTER(C.c_int32))[i] for i in range(d._desc.n_seqs)]
        assert alts == [0, 0, 1]
    finally:
        d.close()
