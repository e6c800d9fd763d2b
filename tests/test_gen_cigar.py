"""CIGAR generation (bm2_gen_cigar, host today, the seam of the device kernel of SURVEY.md 8(f)2) against the reference's own
bwa_gen_cigar2 called through oracle/_ref/refdump cigar: noisy copies of reference windows on both strands, with indels, bands
from 0 to 40, clamped and strand-bridging ranges (the NULL return).  Known answers: score, NM, every CIGAR op and the MD string."""
import subprocess

import numpy as np
import pytest

import bm2
from helpers import ref_binary
from tools import synth
import helpers  # noqa: E402


def _revcomp(a):
    b = a[::-1].copy()
    m = b < 4
    b[m] = 3 - b[m]
    return b


def make_tasks(ctg, seed, n):
    """(query, rb, re, w) tasks: noisy copies of reference windows on both strands, with indels, several bands, plus two NULL cases."""
    genome = np.concatenate(ctg)
    l_pac = len(genome)
    rng = np.random.default_rng(seed)
    tasks = []
    for i in range(n):
        ln = int(rng.integers(20, 400))
        p = int(rng.integers(0, l_pac - ln - 1))
        piece = genome[p:p + ln].copy()
        piece[piece > 3] = int(rng.integers(0, 4))                # the index replaces N by random bases; the query keeps a real base
        q = piece.copy()
        mut = rng.random(ln) < rng.choice([0.0, 0.02, 0.08])
        q[mut] = (q[mut] + 1 + rng.integers(0, 3, size=int(mut.sum()))) % 4
        for _ in range(int(rng.integers(0, 3))):                 # indels
            c = int(rng.integers(3, max(len(q) - 3, 4)))
            if rng.random() < 0.5: q = np.delete(q, slice(c, c + int(rng.integers(1, 5))))
            else: q = np.insert(q, c, rng.integers(0, 4, size=int(rng.integers(1, 5))))
        if rng.random() < 0.03: q[int(rng.integers(0, len(q)))] = 4
        rb, re_ = p, p + ln
        if rng.random() < 0.5:                                   # the same hit on the reverse strand
            q = _revcomp(q.astype(np.uint8)); rb, re_ = 2 * l_pac - (p + ln), 2 * l_pac - p
        tasks.append((q.astype(np.uint8), rb, re_, int(rng.choice([0, 1, 3, 10, 40]))))
    tasks.append((tasks[0][0], l_pac - 50, l_pac + 50, 5))       # bridges the two strands: NULL
    tasks.append((tasks[1][0], 2 * l_pac - 30, 2 * l_pac + 40, 5))   # runs past the end: clamped, NULL
    return tasks


@pytest.mark.parametrize("args,kw", [([], {}), (["-A", "2", "-B", "5", "-O", "7,8", "-E", "2,1"], dict(a=2, b=5, o_del=7, o_ins=8, e_del=2, e_ins=1))])
def test_gen_cigar_matches_reference(tmp_path, args, kw):
    exe, dump = ref_binary(), ref_binary("refdump")
    if exe is None or dump is None:
        helpers.no_checker("oracle/_ref not built (make -C oracle ref)")
    names, ctg, alts = synth.make_genome(17, [120000, 50000], alt_contigs=0, n_repeat_families=3, repeat_len=(200, 1500), copies=(3, 10),
                                         divergence=(0.0, 0.05), n_gaps=2, gap_len=(30, 200))
    fa = str(tmp_path / "g.fa")
    synth.write_fasta(fa, names, ctg)
    subprocess.check_call([exe, "index", fa], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    tasks = make_tasks(ctg, 3 + len(args), 1200)
    tf, of = str(tmp_path / "tasks.txt"), str(tmp_path / "out.bin")
    with open(tf, "w") as f:
        for q, rb, re_, w in tasks:
            f.write("%d %d %d %s\n" % (w, rb, re_, "".join("ACGTN"[c] for c in q)))
    subprocess.check_call([dump] + args + ["cigar", fa, tf, of], stderr=subprocess.DEVNULL)
    raw = open(of, "rb").read()
    exp, p = [], 0
    while p < len(raw):
        sc, nc, nm = np.frombuffer(raw, "<i4", 3, p); p += 12
        if nc < 0:
            exp.append((int(sc), int(nm), None, b"")); continue
        ops = [int(x) for x in np.frombuffer(raw, "<u4", nc, p)]; p += 4 * nc
        e = raw.index(b"\0", p); md = raw[p:e]; p += (e - p + 1 + 3) & ~3
        exp.append((int(sc), int(nm), ops, md))
    got = bm2.gen_cigar(fa, bm2.default_opt(**kw), tasks)
    assert len(exp) == len(got) == len(tasks)
    for i, (x, y) in enumerate(zip(exp, got)):
        if x[2] is None:
            assert y[2] is None, "task %d: the reference returns NULL" % i
        else:
            assert x == y, "task %d (w %d, %d..%d): ref %s ours %s" % (i, tasks[i][3], tasks[i][1], tasks[i][2], x, y)
    assert sum(1 for x in exp if x[2] and len(x[2]) > 1) > 300 and sum(1 for x in exp if x[2] is None) >= 2
