"""Sharding of independent chunks/streams over one process per GPU.

Chunks are independent .lzma streams, so ranks own contiguous chunk ranges and the
data path needs no collective.  The only exchange is an all-gather of the per-chunk
compressed sizes: from it every rank derives where its streams go in the shared
container (the concatenation of all streams in chunk order), so ranks can write
their part without further coordination.  The collective runs over whatever
backend the process group was created with (RCCL on the GPUs, gloo in CPU tests).
"""
import torch


def shard_range(n_items: int, rank: int, world: int):
    """Contiguous [begin, end) of `n_items` owned by `rank`; sizes differ by at most one."""
    if world < 1 or not 0 <= rank < world:
        raise ValueError(f"bad rank {rank} of {world}")
    base, extra = divmod(int(n_items), world)
    begin = rank * base + min(rank, extra)
    return begin, begin + base + (1 if rank < extra else 0)


def shard_by_bytes(sizes, rank: int, world: int):
    """Contiguous [begin, end) of the items owned by `rank`, balanced by bytes instead of by
    count: item i goes to the rank whose share of the total holds the item's first byte
    (rank r's share is bytes [r * total / world, (r + 1) * total / world)).  Every rank
    computes the same cut from the same `sizes`; ranges are contiguous and cover everything.
    Used for the decode batch, whose stream sizes differ by a factor of 16."""
    import numpy as np
    if world < 1 or not 0 <= rank < world:
        raise ValueError(f"bad rank {rank} of {world}")
    sizes = np.asarray(sizes, dtype=np.int64)
    n = int(sizes.size)
    if n == 0:
        return 0, 0
    starts = np.cumsum(sizes) - sizes  # first byte of every item
    total = int(starts[-1] + sizes[-1])
    if total == 0:
        return shard_range(n, rank, world)
    # ceil(r * total / world) in exact integer arithmetic
    lo = (rank * total + world - 1) // world
    hi = ((rank + 1) * total + world - 1) // world
    begin = int(np.searchsorted(starts, lo, side="left"))
    end = n if rank == world - 1 else int(np.searchsorted(starts, hi, side="left"))
    if rank == 0:
        begin = 0
    return begin, max(begin, end)


def shard_counts(n_items: int, world: int):
    return [shard_range(n_items, r, world)[1] - shard_range(n_items, r, world)[0] for r in range(world)]


def gather_sizes(local_sizes: torch.Tensor, n_items: int, rank: int, world: int, group=None):
    """All ranks' per-chunk sizes in chunk order (int64 tensor of n_items on local_sizes' device).

    local_sizes holds the sizes of this rank's shard_range(n_items, rank, world).
    """
    return gather_ragged(local_sizes, shard_counts(n_items, world), rank, world, group)


def gather_ragged(local_sizes: torch.Tensor, counts, rank: int, world: int, group=None):
    """All ranks' values in rank order, rank r contributing counts[r] of them (every rank passes
    the same `counts`): one all-gather, padded to the widest share when the shares differ.
    Used with shard_by_bytes, whose shares differ in length by design."""
    local_sizes = local_sizes.to(torch.int64)
    import torch.distributed as dist
    # (a single rank inside a process group still goes through the collective: that is how the
    # exchange is exercised on a box with one device)
    if world == 1 and not (dist.is_available() and dist.is_initialized()):
        return local_sizes
    counts = [int(c) for c in counts]
    if len(counts) != world:
        raise ValueError(f"{len(counts)} counts for {world} ranks")
    if local_sizes.numel() != counts[rank]:
        raise ValueError(f"rank {rank} holds {local_sizes.numel()} sizes, its shard has {counts[rank]}")
    width = max(counts)
    if all(c == width for c in counts):
        out = torch.empty(world * width, dtype=torch.int64, device=local_sizes.device)
        dist.all_gather_into_tensor(out, local_sizes.contiguous(), group=group)
        return out
    # ragged shards: pad to the widest, gather, drop the padding
    padded = torch.zeros(width, dtype=torch.int64, device=local_sizes.device)
    padded[: counts[rank]] = local_sizes
    out = torch.empty(world * width, dtype=torch.int64, device=local_sizes.device)
    dist.all_gather_into_tensor(out, padded, group=group)
    return torch.cat([out[r * width: r * width + counts[r]] for r in range(world)])


def container_offsets(all_sizes: torch.Tensor):
    """Exclusive prefix sum: offsets[i] = where stream i starts; offsets[-1] = container size."""
    off = torch.zeros(all_sizes.numel() + 1, dtype=torch.int64, device=all_sizes.device)
    torch.cumsum(all_sizes, 0, out=off[1:])
    return off
