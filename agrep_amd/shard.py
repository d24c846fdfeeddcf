"""Sharding of a corpus across ranks (one process per GPU) and the -c / -l aggregate.

Records are independent (every engine resets at a delimiter: asearch.c:175-196,
sgrep.c:1179-1181), so the data path needs no collective: rank r scans its own byte range,
cut so that a record belongs to the range containing its first byte, and the only exchange is
the reduction of the per-rank counts (SUM for -c, MAX for the per-file -l hit flags) over
torch.distributed -- backend "nccl" (= RCCL over xGMI) on GPUs, "gloo" in the CPU tests.
"""
import numpy as np


def shard_pages(total_pages, world, rank):
    """Page range [first, first + count) of `rank` for the synthetic corpus (4 KiB pages end
    with a delimiter, so page boundaries are record boundaries)."""
    base, extra = divmod(total_pages, world)
    first = rank * base + min(rank, extra)
    return first, base + (1 if rank < extra else 0)


def record_cuts(buf, world, delim=10):
    """Byte offsets c[0]=0 <= c[1] <= ... <= c[world]=len(buf): range r = [c[r], c[r+1]).
    Each nominal cut is moved forward to just after the next delimiter, so every record lies
    in exactly one range (the one holding its first byte)."""
    a = np.frombuffer(buf, dtype=np.uint8) if not isinstance(buf, np.ndarray) else buf
    n = a.size
    cuts = [0]
    for r in range(1, world):
        nominal = max(cuts[-1], (n * r) // world)
        if nominal >= n:
            cuts.append(n)
            continue
        # nominal == 0 or the byte before it is a delimiter: already a record start
        if nominal == 0 or a[nominal - 1] == delim:
            cuts.append(nominal)
            continue
        nxt = np.flatnonzero(a[nominal:] == delim)
        cuts.append(n if nxt.size == 0 else nominal + int(nxt[0]) + 1)
    cuts.append(n)
    return cuts


def reduce_counts(n_matched, n_records=0, device=None, group=None):
    """SUM of the per-rank counts over all ranks -> (matched, records) on every rank."""
    import torch
    import torch.distributed as dist
    t = torch.tensor([int(n_matched), int(n_records)], dtype=torch.int64, device=device)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    m, r = t.tolist()                           # one device-to-host round trip
    return int(m), int(r)


def reduce_file_hits(hits, device=None, group=None):
    """-l: elementwise MAX of the per-file hit flags (a file is listed if any rank hit it)."""
    import torch
    import torch.distributed as dist
    t = torch.tensor([int(bool(h)) for h in hits], dtype=torch.int32, device=device)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return [bool(x) for x in t.tolist()]
