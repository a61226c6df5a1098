"""Multi-GPU = batch sharding.  Streams are independent (one reference instance per stream shares nothing:
signalsmith-stretch.h:494-529), so the batch is split into contiguous ranges, one process per GPU, and the data path
needs NO collective.  torch.distributed (RCCL on GPUs, gloo in the CPU tests) is only used by callers for barriers,
the max-over-ranks clock and, optionally, gathering results."""


def shard_range(total_streams, rank, world):
    """Contiguous, balanced partition: the first (total % world) ranks get one extra stream."""
    base, extra = divmod(total_streams, world)
    start = rank*base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def shard_sizes(total_streams, world):
    return [shard_range(total_streams, r, world)[1] - shard_range(total_streams, r, world)[0] for r in range(world)]
