"""Multi-GPU sharding of the loop-closure hot path.

One process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI on the GPUs, "gloo" in
the CPU tests).  The reference has no collective at all; what shards here:
  * descriptor generation : scans are split contiguously over ranks, no communication;
  * descriptor database   : every rank contributes the descriptors it just built with ONE
                            all-gather (ragged shards are padded to the longest), after which
                            every rank holds the whole database, exactly like the per-robot
                            Python lists of RING_ros/main_RING.py:285-289 but device resident;
  * candidate scoring     : the (query, candidate) pair list / the GICP pair list is split
                            contiguously; only the tiny result rows are gathered.
No kernel contains an exchange step.
"""
import torch
import torch.distributed as dist


def shard_bounds(n_items, rank, world):
    """Contiguous, balanced split: the first n_items % world ranks get one extra item."""
    base, extra = divmod(int(n_items), int(world))
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard_sizes(n_items, world):
    return [shard_bounds(n_items, r, world)[1] - shard_bounds(n_items, r, world)[0] for r in range(world)]


def allgather_ragged(local, group=None):
    """All-gather of per-rank tensors whose leading dimension differs.  Returns the
    concatenation in rank order (every rank gets the same tensor)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return local
    world = dist.get_world_size(group)
    n = torch.tensor([local.shape[0]], dtype=torch.int64, device=local.device)
    counts = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(counts, n, group=group)
    counts = [int(c.item()) for c in counts]
    longest = max(counts)
    if longest == 0:
        return local
    padded = local
    if local.shape[0] < longest:
        pad = torch.zeros((longest - local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        padded = torch.cat([local, pad], 0)
    out = torch.empty((world * longest,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, padded.contiguous(), group=group)
    if all(c == longest for c in counts):
        return out
    return torch.cat([out[r * longest: r * longest + counts[r]] for r in range(world)], 0)


def sharded_sweep(queries, local_db, sweep_fn, group=None):
    """Score `queries` (same on every rank) against the database whose rows are spread over the
    ranks.  Alternative to gathering the database: only the [Q, n_local] result rows travel.
    sweep_fn(queries, db) -> (dist [Q,n], angle [Q,n]).  Returns full-width (dist, angle)."""
    d, a = sweep_fn(queries, local_db) if local_db.shape[0] else (
        torch.zeros((queries.shape[0], 0), dtype=torch.float32, device=queries.device),
        torch.zeros((queries.shape[0], 0), dtype=torch.int32, device=queries.device))
    d_all = allgather_ragged(d.t().contiguous(), group).t().contiguous()
    a_all = allgather_ragged(a.t().contiguous(), group).t().contiguous()
    return d_all, a_all
