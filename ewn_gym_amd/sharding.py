"""Lane sharding across GPUs: games are independent, so the env/search path shards with
no exchange step.  GPU g of G owns the contiguous global lanes
[g*N/G, (g+1)*N/G); seeds and Philox streams follow the GLOBAL lane id, so results do
not depend on G.  The only collectives are scalar counters (steps, wins) for reporting."""
import torch


def lane_range(n_total, world_size, rank):
    if n_total % world_size:
        raise ValueError("n_total must be divisible by world_size")
    per = n_total // world_size
    return rank * per, (rank + 1) * per


def lane_seeds(lo, hi, base_seed=9487):
    """seed of global lane i = base_seed + i (reference default seed 9487, envs/ewn.py:37), as uint32 bit patterns"""
    return ((torch.arange(lo, hi, dtype=torch.int64) + base_seed) & 0xFFFFFFFF).to(torch.int32)


def all_reduce_counters(counters, group=None):
    """Sum a small int64 tensor of counters (episodes, wins, steps) over ranks; identity at world_size 1."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(counters, op=dist.ReduceOp.SUM, group=group)
    return counters
