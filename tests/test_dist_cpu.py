"""N>1 plumbing of bench.py on CPU: world_size 2 over gloo (rendezvous on 127.0.0.1), no GPU.
Checks the barrier / max-over-ranks timing protocol and that shards are disjoint, block-aligned and cover the chunk."""
import os
import socket
import sys

import numpy as np
import torch.multiprocessing as mp

from conftest import ROOT


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
