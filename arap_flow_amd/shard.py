"""Frame sharding across GPUs: one process per GPU, frames dealt round-robin, no collective on the data
path.  Reference: para_gen.py:441-445,560-567 (a queue of GPU ids; batches of list-file lines go to whichever
GPU is free; one child process per GPU with CUDA_VISIBLE_DEVICES set, :190).  With equal-cost frames the
static round-robin below is equivalent to that dynamic queue."""
import os


def shard_indices(n_items, rank, world):
    """indices of the items rank `rank` of `world` processes handles: rank, rank + world, ..."""
    if not (0 <= rank < world):
        raise ValueError("rank %d outside world %d" % (rank, world))
    return list(range(rank, n_items, world))


def shard_lines(lines, rank, world):
    return [lines[i] for i in shard_indices(len(lines), rank, world)]


def dist_env():
    """(rank, world, local_rank) from the torch.distributed.run environment; (0, 1, 0) when absent."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")),
            int(os.environ.get("LOCAL_RANK", "0")))


def max_over_ranks(seconds, dist=None, device="cpu"):
    """job time = slowest rank (bench contract).  `dist` is torch.distributed or None."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return float(seconds)
    import torch
    t = torch.tensor([float(seconds)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
