"""SURVEY.md 8(e), in-chunk sharding: ONE chunk cut at multiples of 512 reads over G contexts (bm2_chunk_hits_sharded), hits gathered
in read order, ONE mem_pestat / pairing over the whole chunk.  The SAM must not depend on G and must equal the reference's, for a
chunk whose size is not a multiple of 1024 (so that the parts are uneven and the last one is ragged).
gpu: two and three contexts sharing the replica of one device (on a multi-GPU node tools/bm2_mem.py --contexts takes one per GPU).
Without a GPU: the same through the host emulator, small."""
import os
import subprocess
import sys

import pytest

from helpers import ref_binary
from tools import synth
import helpers  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r'''
import sys, subprocess
sys.path.insert(0, %(root)r); sys.path.insert(0, %(root)r + "/tests"); sys.path.insert(0, %(root)r + "/bwa-mem2_amd")
import bm2
if %(lib)r:
    bm2.LIB_PATH = %(lib)r
from tools import bm2_mem
fa, f1, f2, out = %(fa)r, %(f1)r, %(f2)r, %(out)r
texts = []
for g in %(gs)r:
    bm2_mem.run(fa, [f1, f2], %(K)d, out + str(g), contexts=g, device_tail=True)
    texts.append(open(out + str(g), "rb").read())
assert all(t == texts[0] for t in texts), "the SAM depends on the number of contexts"
ref = subprocess.run([%(exe)r, "mem", "-t", "2", "-K", str(%(K)d), fa, f1, f2], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, check=True).stdout
strip = lambda t: [l for l in t.split(b"\n") if l and not l.startswith(b"@")]
a, b = strip(ref), strip(texts[0])
for i, (x, y) in enumerate(zip(a, b)):
    assert x == y, (i, x[:200], y[:200])
assert len(a) == len(b) and len(a) >= %(nmin)d
print("ok", len(a))
'''


def _case(d, n_pairs):
    names, ctg, alts = synth.make_genome(57, [120000, 50000], alt_contigs=1, alt_len=3000, n_repeat_families=4, repeat_len=(200, 1500),
                                         copies=(3, 15), divergence=(0.0, 0.05))
    fa = os.path.join(d, "g.fa")
    synth.write_fasta(fa, names, ctg)
    synth.write_alt(fa + ".alt", alts)
    subprocess.check_call([ref_binary(), "index", fa], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    r1, r2 = synth.make_reads_pe(58, ctg, n_pairs, L=100, sub_rate=0.01, indel_frac=0.1, random_frac=0.01)
    f1, f2 = os.path.join(d, "r1.fq"), os.path.join(d, "r2.fq")
    synth.write_fastq(f1, r1, suffix="/1"); synth.write_fastq(f2, r2, suffix="/2")
    return fa, f1, f2


def _run(d, lib, n_pairs, gs, K):
    if ref_binary() is None:
        helpers.no_checker("oracle/_ref reference binary not present")
    fa, f1, f2 = _case(d, n_pairs)
    script = SCRIPT % dict(root=ROOT, lib=lib, fa=fa, f1=f1, f2=f2, out=os.path.join(d, "out.sam"), gs=gs, K=K, exe=ref_binary(), nmin=2 * n_pairs)
    p = subprocess.run([sys.executable, "-c", script], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=2400)
    assert p.returncode == 0 and p.stdout.startswith(b"ok"), p.stderr.decode()[-2000:]


def test_sharded_chunk_on_the_emulator(tmp_path, emu_lib):
    lib = emu_lib
    # 2 x 600 reads in one chunk: 3 blocks of 512 -> parts of 512 / 688 reads (2 contexts)
    _run(str(tmp_path), lib, 600, [1, 2], 10 ** 9)


@pytest.mark.gpu
def test_sharded_chunk_on_the_gpu(tmp_path):
    # 2 x 1537 reads per chunk of K bases (not a multiple of 1024), two chunks; 1, 2 and 3 contexts on one device
    _run(str(tmp_path), "", 3074, [1, 2, 3], 1537 * 2 * 100)
