"""Multi-process plumbing of bench.py: one process per GPU, reads sharded across ranks, NO collective on the data path.
torch.distributed is used for three things only: the rendezvous, the barriers around the timed region, and the
max-over-ranks of its duration.  Kept separate so the N>1 logic is testable on CPU with the gloo backend."""
import os

import numpy as np


def env_rank():
    return int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("LOCAL_RANK", 0))


def init(backend, world, device_id=None):
    import torch.distributed as dist
    if world > 1 and not dist.is_initialized():
        import datetime
        kw = {"timeout": datetime.timedelta(minutes=60)}     # rank 0 may spend minutes building the synthetic index behind a barrier
        if device_id is not None:
            kw["device_id"] = device_id
        dist.init_process_group(backend, **kw)


def barrier(world):
    if world > 1:
        import torch.distributed as dist
        dist.barrier()


def max_over_ranks(value, world, device="cpu"):
    if world <= 1:
        return float(value)
    import torch
    import torch.distributed as dist
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def shard_seed(base_seed, rank):
    """Weak scaling: every rank aligns its own synthetic chunk (same size, different reads)."""
    return base_seed + 1000 + rank


def shard_bounds(n_reads, world, block=512):
    """Strong-scaling split of ONE chunk over `world` GPUs: contiguous ranges aligned to the 512-read kt_for block
    (the only cross-read rule of the hot path, bwamem.cpp:834, is per block; mates stay together since 512 is even)."""
    blocks = (n_reads + block - 1) // block
    b = [min(n_reads, (blocks * i // world) * block) for i in range(world + 1)]
    b[-1] = n_reads
    return b


def finish(world):
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()
