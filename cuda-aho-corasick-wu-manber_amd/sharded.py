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
        if _on_gloo() and local_count.is_cuda:  # rehearsal of the N > 1 flow on one card (bench.py --share-device)
            host = local_count.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM)
            local_count.copy_(host)
        else:
            dist.all_reduce(local_count, op=dist.ReduceOp.SUM)
    return local_count


def reduce_count_async(local_count):
    """The same sum, not waited for: returns a work handle (None with one rank).  Under RCCL the all-reduce runs on
    the communicator's own stream behind the kernels that produced `local_count`, so the NEXT scans of the caller's
    stream overlap it; the caller keeps `local_count` untouched until `finish(handle)`."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        if _on_gloo() and local_count.is_cuda:
            reduce_count(local_count)
            return None
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
        if _on_gloo():
            local_counts = local_counts.cpu()
        parts = [torch.zeros_like(local_counts) for _ in range(dist.get_world_size())]
        dist.all_gather(parts, local_counts)
        return torch.stack([p.cpu() for p in parts])
    return local_counts.cpu().unsqueeze(0)


# ---------------------------------------------------------------------------------------------------------
# The 32 GB configurations (BASELINE configs[3], configs[4]) over N ranks: the helpers bench.py runs on the
# GPUs and tests/test_distributed_cpu.py runs over gloo with the CPU lane emulator as the scanner.
def _on_gloo():
    return dist.is_available() and dist.is_initialized() and dist.get_backend() == "gloo"


def world_size():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def shard_plan(n_total, world, rank, lengths):
    """This rank's byte range of an n_total-byte text for every pattern length of a sweep (main.c:467-477 with the
    TRUE length of the last range): -> (begin, bytes to keep resident = range + longest halo, {m: scan length})."""
    spans = {m: shard_for_rank(n_total, world, rank, m) for m in lengths}
    begin = min(b for b, _ in spans.values())
    assert all(b == begin for b, _ in spans.values())
    resident = max(e for _, e in spans.values()) - begin
    return begin, resident, {m: e - b for m, (b, e) in spans.items()}


def verify_slices(n, m, budget, tail=64 << 20):
    """Which part of an n-byte shard the CPU checker recounts: all of it when it fits `budget` bytes (0 = no
    limit), else its head and its last `tail` bytes -- the end with the m-1 halo, where a sharding error would
    show.  -> [(offset, length)], offsets 16-byte aligned (the scan entry points want aligned text)."""
    if budget <= 0 or n <= budget:
        return [(0, n)]
    tail = min(tail, budget // 2)
    off = (n - tail) & ~15
    return [(0, budget - tail), (off, n - off)]


def gather_objects(obj):
    """every rank's `obj` (picklable), in rank order, on every rank; not on a timed path"""
    if world_size() > 1:
        out = [None] * dist.get_world_size()
        dist.all_gather_object(out, obj)
        return out
    return [obj]


def summarize_shard_runs(records, hbm_peak_gbs):
    """`records` = one {"n": scanned bytes, "ms": kernel ms, "matches": count} per rank for ONE pattern set ->
    the line's object for it: the job's rate is all bytes over the SLOWEST device's time (the ranks run side by
    side and the count is complete when the last one is), per-GPU fractions of the HBM peak beside it."""
    worst = max(r["ms"] for r in records)
    total = sum(r["n"] for r in records)
    gbs = total / (worst * 1e-3) / 1e9
    return dict(kernel_ms=round(worst, 4), GBps=round(gbs, 1), Gbit_s=round(8 * gbs, 1),
                hbm_frac=round(min(r["n"] / (r["ms"] * 1e-3) / 1e9 / hbm_peak_gbs for r in records), 4),
                per_gpu_ms=[round(r["ms"], 4) for r in records],
                per_gpu_hbm_frac=[round(r["n"] / (r["ms"] * 1e-3) / 1e9 / hbm_peak_gbs, 4) for r in records],
                matches=sum(r["matches"] for r in records), per_gpu_matches=[r["matches"] for r in records])


def merge_verified(per_rank):
    """`per_rank` = every rank's {name: {"gpu": [...], "cpu": [...], "slices": [...], ...}} -> one dict with N
    per-GPU entries per name and `equal`; a name missing on a rank counts as a failure."""
    names = []
    for d in per_rank:
        for k in d:
            if k not in names:
                names.append(k)
    out = {}
    for name in names:
        per_gpu = [d.get(name) for d in per_rank]
        ok = all(v is not None and v["gpu"] == v["cpu"] for v in per_gpu)
        out[name] = dict(equal=ok, per_gpu=per_gpu)
    return out, all(v["equal"] for v in out.values())


_barrier_calls = {}


def host_barrier(tag, timeout_s=1800.0):
    """A barrier that keeps the GPUs idle: ranks meet in the c10d store (TCP), not in a collective kernel -- an
    RCCL barrier spins on every device, which would sit on the CUs rank 0 is about to measure from one process
    (bench.py's smh_multi leg).  Falls back to dist.barrier() when the store is not reachable.  A tag may be used
    any number of times: every call of every rank appends its own call count, so the k-th call meets the k-th."""
    import time
    if world_size() == 1:
        return
    _barrier_calls[tag] = _barrier_calls.get(tag, 0) + 1
    tag = "%s#%d" % (tag, _barrier_calls[tag])
    try:
        store = dist.distributed_c10d._get_default_store()
        store.add(tag, 1)
        deadline = time.time() + timeout_s
        while store.add(tag, 0) < dist.get_world_size():
            if time.time() > deadline:
                raise RuntimeError("host_barrier(%s): timed out" % tag)
            time.sleep(0.02)
        # rank 0 may be the process that SERVES the store (init_method tcp://, mp.spawn): it must not go on -- and perhaps tear the
        # process group down -- while another rank is still inside a poll above ("Connection reset by peer", seen once in a few
        # runs of tests/test_distributed_cpu.py).  So every rank signs out and rank 0 leaves last.
        store.add(tag + "/left", 1)
        while dist.get_rank() == 0 and store.add(tag + "/left", 0) < dist.get_world_size():
            if time.time() > deadline:
                raise RuntimeError("host_barrier(%s): timed out waiting for the ranks to leave" % tag)
            time.sleep(0.005)
    except (AttributeError, NotImplementedError):
        dist.barrier()
