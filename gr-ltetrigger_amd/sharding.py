"""Channel sharding across ranks: independent capture channels, no data-path collective.

Rank r of `world` owns the global channels [r*per_rank, (r+1)*per_rank) (channel c ->
GPU c // 64).  torch.distributed is used only to line the ranks up (barrier), to take the slowest
rank's wall time and to sum the per-rank match counters -- over gloo on host tensors: the design has no
collective on the data path and therefore no use for RCCL."""


def channel_range(rank, world, per_rank):
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    return range(rank * per_rank, (rank + 1) * per_rank)


def owner_of(channel, per_rank):
    return channel // per_rank


def max_over_ranks(dist, torch, value, device):
    t = torch.tensor([float(value)], device=device, dtype=torch.float64)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def min_over_ranks(dist, torch, value, device):
    t = torch.tensor([float(value)], device=device, dtype=torch.float64)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return float(t.item())


def sum_over_ranks(dist, torch, values, device):
    t = torch.tensor([float(v) for v in values], device=device, dtype=torch.float64)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return [float(v) for v in t.tolist()]


def aggregate_msamp_per_s(world, per_rank, n_samples, steps, elapsed_max):
    """Whole-job throughput: every rank processed per_rank * n_samples * steps samples in elapsed_max."""
    return float(world) * per_rank * n_samples * steps / elapsed_max / 1e6
