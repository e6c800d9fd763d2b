"""Odd inputs through the whole chain (FASTA/FASTQ text -> chunks as `mem -K` -> [oracle in place of the device] -> host tail -> SAM)
against the compiled reference, byte for byte: empty reads, reads shorter than a seed, all-N reads, homopolymers and dinucleotide
runs, chimeras, lower case, FASTA records among FASTQ ones, tabs / comments / 200-character names, `/1` suffixes, unpaired junk as
a mate, chunk sizes that cut the input into many insert-size models.  (60 seeds of this generator were run when it was written;
the two kept here are a regression guard.)"""
import os
import subprocess
import sys

import numpy as np
import pytest

from helpers import oracle_regs_fn, ref_binary
from tools import bm2_mem, synth
import helpers  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _rc(r):
    return np.where(r < 4, 3 - r, 4)[::-1]


def _odd_read(rng, G):
    k = rng.integers(0, 12)
    L = int(rng.choice([0, 1, 5, 18, 19, 20, 30, 50, 100, 150, 151, 250, 400]))
    if k == 0:
        return rng.integers(0, 4, L).astype(np.uint8)
    if k == 1:
        return np.full(L, 4, np.uint8)
    s = int(rng.integers(0, len(G) - L - 1)) if L else 0
    r = G[s:s + L].copy()
    if k == 2:
        r[rng.random(L) < 0.2] = 4
    if k == 3 and L > 40:
        r = np.concatenate([r[:L // 2], G[(s + 5000) % (len(G) - L):][:L - L // 2]])
    if k == 4:
        r = np.tile(np.array([0, 1], np.uint8), L // 2 + 1)[:L]
    if k == 5:
        r = np.zeros(L, np.uint8)
    if k == 6 and L > 30:
        r = np.concatenate([r[:L // 2], r[L // 2 + 7:], G[s + L:s + L + 7]])[:L]
    if k == 7 and L > 30:
        r = np.concatenate([r[:L // 2], rng.integers(0, 4, 9).astype(np.uint8), r[L // 2:]])[:L]
    m = rng.random(len(r)) < rng.choice([0, 0.01, 0.05, 0.12])
    r = r.copy()
    r[m & (r < 4)] = (r[m & (r < 4)] + 1) % 4
    if rng.random() < 0.5:
        r = _rc(r)
    return r.astype(np.uint8)


def _name(rng, i):
    k = rng.integers(0, 5)
    return [b"r%d" % i, b"r%d/1" % i, b"read_with_a_really_long_name_%d_" % i + b"x" * 200, b"r%d comment BC:Z:ACGT" % i, b"r%d\tXX:i:1" % i][k]


def _write(rng, path, reads, names):
    with open(path, "wb") as f:
        for i, r in enumerate(reads):
            s = bytes(b"ACGTN"[c] for c in r)
            if rng.random() < 0.1:
                s = s.lower()
            if rng.random() < 0.15:
                f.write(b">" + names[i] + b"\n" + s + b"\n")
            else:
                f.write(b"@" + names[i] + b"\n" + s + b"\n+\n" + bytes(rng.integers(33, 74, size=len(r), dtype=np.uint8)) + b"\n")


@pytest.mark.parametrize("seed,paired", [(5, False), (6, True)])
def test_odd_inputs_fastq_to_sam(tmp_path, seed, paired):
    exe = ref_binary()
    if exe is None:
        helpers.no_checker("oracle/_ref reference binary not present (build it with `make -C oracle ref`)")
    rng = np.random.default_rng(seed)
    names, ctg, alts = synth.make_genome(seed, [80000, 30000, 5000], alt_contigs=1, alt_len=3000, n_repeat_families=5, repeat_len=(100, 1500),
                                         copies=(3, 20), divergence=(0.0, 0.05), n_gaps=2, gap_len=(50, 300))
    fa = str(tmp_path / "g.fa")
    synth.write_fasta(fa, names, ctg)
    synth.write_alt(fa + ".alt", alts)
    subprocess.check_call([exe, "index", fa], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    G = np.concatenate(ctg)
    n = 300
    nm = [_name(rng, i) for i in range(n)]
    files = [str(tmp_path / "a_1.fq")]
    _write(rng, files[0], [_odd_read(rng, G) for _ in range(n)], nm)
    if paired:
        r2 = []
        for _ in range(n):
            if rng.random() < 0.5:
                r2.append(_odd_read(rng, G))
            else:
                s = int(rng.integers(300, len(G) - 300))
                r2.append(_rc(G[s:s + 100]).astype(np.uint8))
        files.append(str(tmp_path / "a_2.fq"))
        _write(rng, files[1], r2, nm)
    K = "3000" if paired else "20000"
    p = subprocess.run([exe, "mem", "-t", "1", "-K", K, fa] + files, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, check=True)
    ref = b"".join(l for l in p.stdout.splitlines(keepends=True) if not l.startswith(b"@PG"))
    out = str(tmp_path / "o.sam")
    bm2_mem.run(fa, files, int(K), out, hits_of=oracle_regs_fn(fa))
    got = open(out, "rb").read()
    if ref != got:
        la, lb = ref.splitlines(), got.splitlines()
        for i, (x, y) in enumerate(zip(la, lb)):
            assert x == y, "line %d\n  ref : %s\n  ours: %s" % (i, x[:500], y[:500])
        assert len(la) == len(lb)
