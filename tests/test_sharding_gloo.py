"""Multi-GPU path on CPU: world_size 2 over gloo.  Games shard by index with no data-path collective; a game's
result depends only on its GLOBAL index (RNG key), so the union of the ranks' outputs equals the single-process
run; the measurement reduction is MAX(time) / SUM(work).  On the CPU the oracle stands in for the engine (tests only); the same
body runs the real engine under `-m gpu` (two ranks = two engines on the one GPU, offset first_global_game)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def play_game(global_idx, seed=11):
    """One TicTacToe self-play game whose RNG is keyed by (seed, global game index)."""
    from fixture_eval import fixture_logits_value
    from oracle import az_oracle as ao
    game = ao.OracleGame("tictactoe")
    rng = np.random.Generator(np.random.Philox(key=[seed, global_idx]))

    def ev(canon):
        logits, v = fixture_logits_value(torch.from_numpy(np.ascontiguousarray(canon))[None], 9, "hash")
        return ao.softmax_det(logits[0].numpy()), float(v[0])
    out = ao.self_play(game, ev, 20, noise_fn=lambda mc: rng.dirichlet([0.3] * 9), uniform_fn=lambda mc: rng.random())
    return out["cells"].tolist(), int(out["winner"])


def _worker(rank, world, port, games_per_rank, q):
    sys.path[:0] = [ROOT, os.path.join(ROOT, "alpha-zero_amd"), os.path.join(ROOT, "tests")]
    from shard import reduce_measurement, shard_range
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    first, last = shard_range(games_per_rank, rank)
    mine = {g: play_game(g) for g in range(first, last)}
    dist.barrier()
    elapsed = 1.0 + rank                               # deterministic stand-in for a per-rank clock
    t, (n_games, plies) = reduce_measurement(elapsed, [len(mine), sum(len(v[0]) for v in mine.values())], dist, "cpu")
    gathered = [None] * world
    dist.all_gather_object(gathered, mine)             # test-side collection only
    if rank == 0:
        merged = {}
        for d in gathered:
            merged.update(d)
        q.put((t, n_games, plies, merged))
    dist.destroy_process_group()


def test_two_ranks_equal_one_process():
    from shard import shard_range, split_games
    world, gpr = 2, 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, world, port, gpr, q)) for r in range(world)]
    for p in procs:
        p.start()
    t, n_games, plies, merged = q.get(timeout=120)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    single = {g: play_game(g) for g in range(world * gpr)}
    assert merged == single                            # sharding does not change any game
    assert t == 2.0 and n_games == world * gpr         # MAX over ranks, SUM of work
    assert plies == sum(len(v[0]) for v in single.values())
    assert [shard_range(2048, r) for r in range(3)] == [(0, 2048), (2048, 4096), (4096, 6144)]
    assert [split_games(10, r, 4) for r in range(4)] == [(0, 3), (3, 6), (6, 8), (8, 10)]


def _engine_rank(rank, games_per_rank, n_sims=40):
    """What one rank of bench.py does for its shard, with the real engine: (cells, winner) per GLOBAL game index."""
    from fixture_eval import fixture_logits_value
    from selfplay import self_play_batch
    from shard import shard_range
    first, last = shard_range(games_per_rank, rank)
    res = self_play_batch("gomoku", lambda x: fixture_logits_value(x, 49, "hash"), games_per_rank, n_sims, size=7, seed=11,
                          first_global_game=first, cache_entries=256, cache_shared=True)
    return {first + i: (r.cells, r.winner) for i, r in enumerate(res)}


@pytest.mark.gpu
def test_two_engine_shards_equal_one_engine():
    """The sharding invariant with the engine itself: rank r plays global games [r G, (r+1) G) with RNG keyed by the global index,
    so two shards (two engines, here on one GPU) produce exactly the games of one engine holding all of them."""
    gpr = 12
    merged = {}
    for rank in range(2):
        merged.update(_engine_rank(rank, gpr))
    single = _engine_rank(0, 2 * gpr)
    assert merged == single and len(single) == 2 * gpr
