"""Byte-range sharding over ranks: one process per GPU, counts summed with one all-reduce.

Mirrors what the reference driver does with MPI (main.c:375-378 shard length,
main.c:464-477 displacements with the m-1 halo, main.c:654-657 MPI_Reduce of the
count), with two differences: every rank works on its TRUE shard length (the
reference passes the padded length, main.c:376,630) and the count is 64-bit.

The data path has no collective: shards are independent; the only exchange is
the 8-byte sum.  With backend "nccl" that is RCCL over xGMI; the CPU tests run
the same code over "gloo".
"""
import torch
import torch.distributed as dist

import smatcher_hip as S


def shard_for_rank(n_total, world_size, rank, m):
    """[begin, end) of this rank's bytes, halo included (smh_shard_range = main.c:467-477)."""
    return S.shard_range(n_total, world_size, rank, m)


def reduce_count(local_count):
    """Sum a per-rank match count over all ranks; `local_count` is a 1-element int64 tensor
    (device tensor under nccl/RCCL, CPU tensor under gloo).  Returns the same tensor."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(local_count, op=dist.ReduceOp.SUM)
    return local_count


def reduce_count_async(local_count):
    """The same sum, not waited for: returns a work handle (None with one rank).  Under RCCL the all-reduce runs on
    the communicator's own stream behind the kernels that produced `local_count`, so the NEXT scans of the caller's
    stream overlap it; the caller keeps `local_count` untouched until `finish(handle)`."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        return dist.all_reduce(local_count, op=dist.ReduceOp.SUM, async_op=True)
    return None


def finish(handles):
    """Wait for the reductions started with reduce_count_async (makes the current stream wait for them)."""
    for h in handles:
        if h is not None:
            h.wait()


def gather_counts(local_counts):
    """Every rank's per-shard counts, for the report (and as a parity check against per-shard CPU counts):
    `local_counts` is an int64 tensor of the same shape on every rank; returns a [world, ...] int64 CPU
    tensor on every rank.  One all-gather of a few bytes; not on the timed path."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        parts = [torch.zeros_like(local_counts) for _ in range(dist.get_world_size())]
        dist.all_gather(parts, local_counts)
        return torch.stack([p.cpu() for p in parts])
    return local_counts.cpu().unsqueeze(0)
