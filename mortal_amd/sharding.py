"""Multi-GPU sharding of the table pool: one process per GPU, disjoint contiguous game ranges, and the single
collective of the data path — the gather of episode returns.

Tables are independent (per-game RNG keyed by (seed, key, kyoku, honba), arena/board.rs:99-107), so nothing is
exchanged per step.  Ranges are multiples of 4 so that the four seat rotations of one seed (one_vs_three.rs:140-156)
stay on one rank.
"""
import numpy as np
import torch
import torch.distributed as dist


def shard_range(n_games, rank, world, group=4):
    """[g0, g1) of `n_games` (a multiple of `group`) owned by `rank`; `group` = games per seed (4 seat rotations in
    OneVsThree, 2 splits in TwoVsTwo), kept together on one rank."""
    assert n_games % group == 0
    sets = n_games // group
    base, rem = divmod(sets, world)
    s0 = rank * base + min(rank, rem)
    s1 = s0 + base + (1 if rank < rem else 0)
    return group * s0, group * s1


def dist_info():
    """(rank, world, backend) of the default process group, (0, 1, None) when torch.distributed is not initialised."""
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size(), dist.get_backend()
    return 0, 1, None


def seeds_for(seed_start, g0, g1):
    """Game g uses (seed_start[0] + g // 4, key) (one_vs_three.rs:140-142)."""
    return [(int(seed_start[0]) + g // 4, int(seed_start[1])) for g in range(g0, g1)]


def gather_returns(local_scores, n_games, group=None, device=None):
    """All ranks pass their [g1-g0, 4] int32 final scores; rank 0 gets the [n_games, 4] array (others None)."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    device = device or torch.device("cpu")
    counts = [shard_range(n_games, r, world) for r in range(world)]
    width = max(g1 - g0 for g0, g1 in counts)
    buf = torch.zeros((width, 4), dtype=torch.int32, device=device)
    buf[: local_scores.shape[0]] = torch.as_tensor(np.asarray(local_scores), dtype=torch.int32, device=device)
    out = [torch.empty_like(buf) for _ in range(world)] if rank == 0 else None
    dist.gather(buf, out, dst=0, group=group)
    if rank != 0:
        return None
    return np.concatenate([out[r][: g1 - g0].cpu().numpy() for r, (g0, g1) in enumerate(counts)], axis=0)


def allreduce_rank_histogram(local_hist, group=None, device=None):
    """Sum of the per-rank challenger rank histograms == py_vs_py's return value (one_vs_three.rs:55-60)."""
    t = torch.as_tensor(list(local_hist), dtype=torch.int64, device=device or torch.device("cpu"))
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return [int(x) for x in t.tolist()]
