"""Multi-GPU layout of self-play: games are independent, so they shard by index with no collective on the
generation path.  One process per GPU; rank r owns global games [r*G, (r+1)*G) and keys its RNG
by the GLOBAL game index, so what a game does is independent of how many GPUs run.  The only collectives are
measurement: a barrier around the timed region, MAX of the elapsed time and SUM of the work counters."""
import os


def env_world():
    """(rank, local_rank, world_size) as torch.distributed.run exports them (1 process = (0, 0, 1))."""
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def shard_range(games_per_rank, rank):
    """Global game indices owned by `rank` (weak scaling: per-GPU work is fixed)."""
    first = rank * games_per_rank
    return first, first + games_per_rank


def split_games(total_games, rank, world):
    """Strong-scaling split of a fixed job: contiguous blocks, the first `total % world` ranks get one more."""
    base, rem = divmod(total_games, world)
    first = rank * base + min(rank, rem)
    return first, first + base + (1 if rank < rem else 0)


def reduce_measurement(elapsed_s, work, dist=None, device=None):
    """max-over-ranks time and summed work.  `work` is a list of numbers; dist = torch.distributed or None."""
    import torch
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return float(elapsed_s), [float(w) for w in work]
    t = torch.tensor([elapsed_s], dtype=torch.float64, device=device)
    w = torch.tensor(list(work), dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.all_reduce(w, op=dist.ReduceOp.SUM)
    return float(t[0]), [float(x) for x in w]


def agree_min(value, dist=None, device=None):
    """The smallest `value` over the ranks (an integer every rank then uses to decide how many collective-bearing steps
    to run).  A quantity that depends on a rank's own games - its replay fill, its number of finished games - must never
    decide the NUMBER of collectives a rank issues: ranks that disagree deadlock in RCCL.  One tiny all-reduce(MIN)."""
    import torch
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return int(value)
    t = torch.tensor([int(value)], dtype=torch.int64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return int(t[0])
