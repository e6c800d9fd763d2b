"""The oracle against the reference (oracle/_ref/refdump) under NON-default scoring: asymmetric gap penalties, -A scaling,
and the Z-drop settings where the reference's int8 / int16 SIMD kernels stop behaving like its scalar kernel (ZSCORE8 /
ZSCORE16, bandedSWA.cpp:268-281: zdrop truncated to the lane width, no multiplication by the gap extension penalty,
evaluated on every row).  Every stage must be byte-identical.  No GPU."""
import subprocess

import numpy as np
import pytest

from helpers import ref_binary
from tools import oracle, refio, synth
import helpers  # noqa: E402

SETS = [
    (["-E", "3,1"], dict(e_del=3, e_ins=1)),
    (["-O", "9,3", "-E", "1,2", "-w", "20", "-L", "2,9", "-d", "30"], dict(o_del=9, o_ins=3, e_del=1, e_ins=2, w=20, pen_clip5=2, pen_clip3=9, zdrop=30)),
    (["-A", "2"], dict(a=2, b=8, o_del=12, e_del=2, o_ins=12, e_ins=2, zdrop=200, pen_clip5=10, pen_clip3=10)),   # -d 200 is -56 in int8 lanes
    (["-d", "0"], dict(zdrop=0)),
    (["-d", "150", "-E", "2,2"], dict(zdrop=150, e_del=2, e_ins=2)),
]


@pytest.mark.parametrize("L", [100, 250])
def test_oracle_matches_reference_under_non_default_scoring(tmp_path, L):
    exe, dump = ref_binary(), ref_binary("refdump")
    if exe is None or dump is None:
        helpers.no_checker("oracle/_ref not built (make -C oracle ref)")
    names, ctg, alts = synth.make_genome(77, [200000, 90000], alt_contigs=1, alt_len=4000, n_repeat_families=6, repeat_len=(200, 2500),
                                         copies=(3, 30), divergence=(0.0, 0.06))
    fa = str(tmp_path / "g.fa")
    synth.write_fasta(fa, names, ctg)
    synth.write_alt(fa + ".alt", alts)
    subprocess.check_call([exe, "index", fa], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    reads = synth.make_reads_se(78, ctg, 1200, L=L, sub_rate=0.04, indel_frac=0.3, random_frac=0.02)
    rt = str(tmp_path / "reads.txt")
    with open(rt, "w") as f:
        for r in reads:
            f.write("".join("ACGTN"[c] for c in r) + "\n")
    enc, off, ln = refio.pack_reads(list(reads))
    ix = oracle.Index(fa)
    try:
        for args, kw in SETS:
            out = str(tmp_path / "dump")
            subprocess.check_call([dump] + args + [fa, rt, out], stderr=subprocess.DEVNULL)
            d = refio.read_dump(out)
            exp = ix.run(enc, off, ln, oracle.default_opt(**kw))
            for t in ("SMEM", "SACOORD", "CHN1", "SEED1", "REGRAW", "REGPRG"):
                assert d[t].tobytes() == exp[t].tobytes(), "%s: stage %s differs" % (" ".join(args), t)
    finally:
        ix.close()


@pytest.mark.parametrize("case", ["L36", "L600_noisy", "long_pacbio"])
def test_oracle_matches_reference_across_read_shapes(tmp_path, case):
    from helpers import ONT2D
    exe, dump = ref_binary(), ref_binary("refdump")
    if exe is None or dump is None:
        helpers.no_checker("oracle/_ref not built (make -C oracle ref)")
    names, ctg, alts = synth.make_genome(101, [300000, 120000, 30000], alt_contigs=2, alt_len=5000, n_repeat_families=10, repeat_len=(100, 4000),
                                         copies=(2, 80), divergence=(0.0, 0.1), n_gaps=6, gap_len=(20, 800))
    fa = str(tmp_path / "g.fa")
    synth.write_fasta(fa, names, ctg)
    synth.write_alt(fa + ".alt", alts)
    subprocess.check_call([exe, "index", fa], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    if case == "L36":
        reads, args, kw = synth.make_reads_se(236, ctg, 2500, L=36, n_frac=0.03), ["-x", "intractg"], dict(b=9, o_del=16, o_ins=16, pen_clip5=5, pen_clip3=5)
    elif case == "L600_noisy":
        reads, args, kw = synth.make_reads_se(800, ctg, 500, L=600, sub_rate=0.06, indel_frac=0.5), [], {}
    else:
        reads, args, kw = synth.make_reads_long(333, ctg, 60, mean_len=3000, max_len=9000, err=0.1), ["-x", "pacbio"], dict(ONT2D, min_seed_len=17, min_chain_weight=40)
    rt = str(tmp_path / "reads.txt")
    with open(rt, "w") as f:
        for r in reads:
            f.write("".join("ACGTN"[c] for c in r) + "\n")
    out = str(tmp_path / "dump")
    subprocess.check_call([dump] + args + [fa, rt, out], stderr=subprocess.DEVNULL)
    d = refio.read_dump(out)
    enc, off, ln = refio.pack_reads(list(reads))
    ix = oracle.Index(fa)
    try:
        exp = ix.run(enc, off, ln, oracle.default_opt(**kw))
    finally:
        ix.close()
    for t in ("SMEM", "SACOORD", "CHN1", "SEED1", "REGRAW", "REGPRG"):
        assert d[t].tobytes() == exp[t].tobytes(), "%s: stage %s differs" % (case, t)
