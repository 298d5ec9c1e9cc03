"""Channel sharding for multi-GPU runs: channels are independent, so GPU g of G owns the
contiguous range [g*C/G, (g+1)*C/G) -- state, parameters, I/Q rows and output rows all live on that GPU.
There is no data-path collective; torch.distributed is only used by callers for barriers / timing."""


def shard_range(n_channels, rank, world):
    """Half-open channel range of `rank` (balanced to within one channel)."""
    lo = (n_channels * rank) // world
    hi = (n_channels * (rank + 1)) // world
    return lo, hi
