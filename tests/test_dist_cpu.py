"""N>1 plumbing of bench.py on CPU: world_size 2 over gloo (rendezvous on 127.0.0.1), no GPU.
Checks the barrier / max-over-ranks timing protocol and that shards are disjoint, block-aligned and cover the chunk."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

from conftest import ROOT
import helpers  # noqa: E402


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from tools import dist_util, synth
    r, w, l = dist_util.env_rank()
    assert (r, w, l) == (rank, world, rank)
    dist_util.init("gloo", world)
    dist_util.barrier(world)
    # each rank "measures" a different duration: every rank must see the max
    t = dist_util.max_over_ranks(1.0 + rank, world)
    # weak-scaling shards: same size, different reads
    names, ctg, _ = synth.make_genome(5, [30000, 20000], alt_contigs=0)
    reads = synth.make_reads_se(dist_util.shard_seed(7, rank), ctg, 256, L=100)
    out.put((rank, t, int(reads.astype(np.int64).sum())))
    dist_util.finish(world)


def test_two_ranks_gloo():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    assert [r for r, _, _ in res] == [0, 1]
    assert all(t == 2.0 for _, t, _ in res)              # max over ranks of (1.0, 2.0)
    assert res[0][2] != res[1][2]                         # different reads per rank


def test_shard_bounds_are_block_aligned_and_cover():
    sys.path.insert(0, ROOT)
    from tools import dist_util
    for n in (0, 1, 511, 512, 513, 100000, 1000003):
        for w in (1, 2, 4, 8):
            b = dist_util.shard_bounds(n, w)
            assert b[0] == 0 and b[-1] == n and all(x <= y for x, y in zip(b, b[1:]))
            assert all(x % 512 == 0 for x in b[:-1])


def _align_worker(rank, world, port, lib, fa, npz, out):
    """one rank of the strong-scaling scheme: its part of ONE chunk through the (emulated) device pipeline, hits gathered on rank 0"""
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "bwa-mem2_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    import bm2
    from tools import dist_util
    bm2.LIB_PATH = lib
    dist_util.init("gloo", world)
    z = np.load(npz)
    enc, off, ln = z["enc"], z["off"], z["ln"]
    b = dist_util.shard_bounds(len(ln), world)
    lo, hi = b[rank], b[rank + 1]
    e0 = int(off[lo]); e1 = int(off[hi - 1] + ln[hi - 1]) if hi > lo else e0
    ctx = bm2.Context(0, fa)
    opt = bm2.default_opt()
    ctx.batch_upload(enc[e0:e1], off[lo:hi] - e0, ln[lo:hi]); ctx.batch_run(opt); ctx.batch_finish(opt)
    aln, aln_off = ctx.batch_download_alnregs()
    ctx.close()
    parts = [None] * world if rank == 0 else None
    dist.gather_object((lo, hi, aln.tobytes(), aln_off.tobytes()), parts, dst=0)          # no collective on the data path: results only
    if rank == 0:
        out.put(parts)
    dist_util.finish(world)


@pytest.mark.parametrize("world,n_pairs", [(2, 300), (8, 2100)], ids=["two_ranks", "eight_ranks"])
def test_ranks_align_one_chunk(tmp_path, emu_lib, world, n_pairs):
    # SURVEY.md 8(e) with processes: ONE chunk cut at multiples of 512 reads over two / EIGHT ranks (gloo), every rank runs the device
    # pipeline (host emulator of the device sources) on its part, rank 0 gathers the hits in read order: equal to the oracle's
    # mem_alnreg_v contents of the whole chunk, i.e. independent of the sharding
    import subprocess
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools", "emu")); sys.path.insert(0, os.path.join(ROOT, "bwa-mem2_amd"))
    import bm2
    from helpers import alnregs_to_recs, ref_binary
    from tools import oracle, refio, synth
    if ref_binary() is None:
        helpers.no_checker("oracle/_ref reference binary not present")
    lib = emu_lib
    names, ctg, _ = synth.make_genome(5, [60000, 30000], alt_contigs=0, n_repeat_families=2, repeat_len=(200, 800), copies=(3, 8))
    fa = str(tmp_path / "g.fa")
    synth.write_fasta(fa, names, ctg)
    subprocess.check_call([ref_binary(), "index", fa], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    r1, r2 = synth.make_reads_pe(9, ctg, n_pairs, L=100)                # 600 reads: parts of 512 and 88; 4200 reads: seven parts of 512 and one of 616
    reads = np.empty((2 * n_pairs, 100), np.uint8); reads[0::2] = r1; reads[1::2] = r2
    enc, off, ln = refio.pack_reads(reads)
    npz = str(tmp_path / "chunk.npz")
    np.savez(npz, enc=enc, off=off, ln=ln)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_align_worker, args=(r, world, port, lib, fa, npz, q)) for r in range(world)]
    for p in ps:
        p.start()
    parts = q.get(timeout=1200)
    for p in ps:
        p.join(120)
        assert p.exitcode == 0
    parts.sort()
    n = 2 * n_pairs
    assert [(lo, hi) for lo, hi, _, _ in parts] == ([(0, 512), (512, 600)] if world == 2 else [(512 * i, 512 * (i + 1)) for i in range(7)] + [(3584, n)])
    aln = np.concatenate([np.frombuffer(a, bm2.ALNREG_DT) for _, _, a, _ in parts])
    aln_off, base = [0], 0
    for lo, hi, a, o in parts:
        o = np.frombuffer(o, np.int64)
        aln_off += list(o[1:] + base); base += int(o[-1])
    ix = oracle.Index(fa); exp = ix.run(enc, off, ln)["REGFIN"]; ix.close()
    got = alnregs_to_recs(aln, np.array(aln_off, np.int64))
    assert len(got) == len(exp) and got.tobytes() == exp.tobytes()
