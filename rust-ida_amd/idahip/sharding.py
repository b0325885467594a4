"""Ensemble sharding across the GPUs of one node: independent IVPs, contiguous block of systems per
rank, no data-path collective. torch.distributed is used only to line the ranks up and to combine the two scalars a
throughput figure needs (max elapsed time, total Newton iterations)."""


def shard_range(rank, world, batch_per_rank):
    """Global system ids [first, first + count) integrated by `rank` (weak scaling: every rank owns batch_per_rank)."""
    if not (0 <= rank < world):
        raise ValueError("rank %d outside world of %d" % (rank, world))
    return rank * batch_per_rank, batch_per_rank


def combine(elapsed_s, newton_iters, dist=None, device=None):
    """-> (max over ranks of elapsed_s, sum over ranks of newton_iters). `dist` is an initialised torch.distributed
    module (nccl = RCCL on the GPU box, gloo in the CPU tests) or None for a single process."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return float(elapsed_s), int(newton_iters)
    import torch
    t = torch.tensor([float(elapsed_s)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    n = torch.tensor([int(newton_iters)], dtype=torch.int64, device=device)
    dist.all_reduce(n, op=dist.ReduceOp.SUM)
    return float(t.item()), int(n.item())
