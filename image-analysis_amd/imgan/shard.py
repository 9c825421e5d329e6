"""Multi-GPU layout of the path: images are independent (basegen.py:96-99 loops over files),
so a batch is cut into contiguous shards, one per rank, with no halo and no per-iteration
exchange.  The only collective is the all-gather of the fixed-width descriptor matrix into
the replicated retrieval index (RCCL over xGMI on GPUs; gloo in the CPU tests)."""
import torch
import torch.distributed as dist


def shard_range(n_items, rank, world):
    """Contiguous shard [lo, hi) of rank `rank`: sizes differ by at most one."""
    if not (0 <= rank < world):
        raise ValueError("rank %d outside world of %d" % (rank, world))
    base, extra = divmod(int(n_items), int(world))
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


class PendingGather:
    """An all-gather in flight (gather_descriptors_async).  result() makes the current stream
    wait for it and returns the [n_total, D] matrix."""

    def __init__(self, out, work=None, sizes=None, width=0):
        self._out, self._work, self._sizes, self._width = out, work, sizes, width

    def result(self):
        if self._work is not None:
            self._work.wait()
            self._work = None
        out, sizes, width = self._out, self._sizes, self._width
        if sizes is None or all(s == width for s in sizes):
            return out
        parts = [out[r * width:r * width + sizes[r]] for r in range(len(sizes))]
        return torch.cat(parts, dim=0)


def gather_descriptors_async(local, n_total=None, group=None):
    """All-gather [n_local, D] descriptor rows of every rank into [n_total, D] (rank order =
    image order, because shards are contiguous) without waiting for it: the collective runs on
    the backend's own stream while the caller enqueues the next batch.  The rows are copied
    first, so the caller may overwrite `local` at once.  Ragged shards are padded to the
    largest shard for the collective and trimmed in result()."""
    if not dist.is_available() or not dist.is_initialized():
        return PendingGather(local)
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    if local.is_cuda and dist.get_backend(group) == "gloo":
        # gloo has no all-gather for device tensors (rehearsals of the N > 1 flow on one GPU)
        return PendingGather(gather_descriptors_async(local.cpu(), n_total, group).result()
                             .to(local.device))
    if n_total is None:
        counts = torch.tensor([local.shape[0]], dtype=torch.int64, device=local.device)
        all_counts = [torch.zeros_like(counts) for _ in range(world)]
        dist.all_gather(all_counts, counts, group=group)
        sizes = [int(c.item()) for c in all_counts]
    else:
        sizes = [shard_range(n_total, r, world)[1] - shard_range(n_total, r, world)[0]
                 for r in range(world)]
    if sizes[rank] != local.shape[0]:
        raise ValueError("rank %d holds %d rows, expected %d" % (rank, local.shape[0], sizes[rank]))
    width = max(sizes)
    padded = torch.zeros((width,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    padded[:local.shape[0]] = local
    out = torch.empty((world * width,) + tuple(local.shape[1:]), dtype=local.dtype,
                      device=local.device)
    work = dist.all_gather_into_tensor(out, padded, group=group, async_op=True)
    return PendingGather(out, work, sizes, width)


def gather_descriptors(local, n_total=None, group=None):
    """Blocking form of gather_descriptors_async."""
    return gather_descriptors_async(local, n_total, group).result()
