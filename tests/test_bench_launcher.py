"""bench.py's multi-rank launch path on the CPU (gloo, stub runner): `python bench.py --gpus N` without
torch.distributed.run spawns its own ranks and prints ONE line with n_gpus = N and work summed over the ranks; a world
size that disagrees with --gpus is a non-zero exit, never a silent one-GPU run.
Also: the collective COUNT of the training leg must not depend on a rank's own replay fill."""
import json
import os
import subprocess
import sys

import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT

BENCH = os.path.join(ROOT, "bench.py")
ARGS = ["--stub-engine", "--steps", "5", "--warmup", "2", "--games", "8", "--sims", "10", "--preroll-cheap", "3", "--preroll-full", "2"]


def _run(extra, env=None):
    e = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, BENCH] + extra + ARGS, env=e, capture_output=True, text=True, timeout=300)


def _line(out):
    return json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])


def test_gpus_2_spawns_two_ranks_and_sums_their_work():
    one = _run(["--gpus", "1"])
    assert one.returncode == 0, one.stderr
    two = _run(["--gpus", "2"])
    assert two.returncode == 0, two.stderr
    a, b = _line(one), _line(two)
    assert a["n_gpus"] == 1 and b["n_gpus"] == 2 and b["data"] == "stub"
    assert len([l for l in two.stdout.splitlines() if l.startswith("{")]) == 1        # rank 0 only
    assert b["plies_in_window"] == 2 * a["plies_in_window"]                           # SUM over ranks (weak scaling)
    # the stub's completions depend on the shard offset: rank 1 differs from rank 0, and both are counted
    per_rank = [sum((first + m) % 7 + 1 for m in range(3 + 2 + 2 + 1, 3 + 2 + 2 + 5 + 1)) for first in (0, 8)]
    assert a["games_finished_in_window"] == per_rank[0] and b["games_finished_in_window"] == sum(per_rank)


def test_world_size_mismatch_is_an_error():
    r = _run(["--gpus", "2"], env={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "refusing" in r.stderr
    r = _run(["--gpus", "1"], env={"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29999"})
    assert r.returncode != 0


def _agree_worker(rank, world, port, q):
    sys.path[:0] = [ROOT, os.path.join(ROOT, "alpha-zero_amd"), os.path.join(ROOT, "tests")]
    import torch
    from shard import agree_min
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    # unequal replay fills: rank 0 holds 5 batches, rank 1 only 2 -> both must run exactly 2 collective-bearing steps
    fill = [5 * 512 + 17, 2 * 512 + 300][rank]
    steps = agree_min(fill // 512, dist, "cpu")
    n_allreduce = 0
    for _ in range(steps):
        t = torch.ones(4)
        dist.all_reduce(t)                    # would hang (and time the test out) if the ranks disagreed on `steps`
        n_allreduce += 1
    # the bench's per-move decision: a step runs only when EVERY rank holds a batch
    ready = [agree_min(1 if f >= 512 else 0, dist, "cpu") for f in ([600, 100][rank], [600, 512][rank])]
    q.put((rank, steps, n_allreduce, ready))
    dist.destroy_process_group()


def test_unequal_replay_fills_agree_on_the_collective_count():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29700 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_agree_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert got == [(0, 2, 2, [0, 1]), (1, 2, 2, [0, 1])]
