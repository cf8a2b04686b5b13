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


def _train_promote_worker(rank, world, port, q):
    """One rank of a world-2 config-5 iteration on the CPU: unequal replay fills -> agreed step count -> Trainer.train with the
    fused gradient bucket all-reduced (gloo) -> promotion into the rank's self-play evaluator IN PLACE."""
    sys.path[:0] = [ROOT, os.path.join(ROOT, "alpha-zero_amd"), os.path.join(ROOT, "tests")]
    import hashlib
    import numpy as np
    import torch
    from pvnet import NetConfig, PolicyValueNet, init_weights
    from shard import agree_min
    from trainer import Trainer
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    cfg = NetConfig(7, 7, 2, 49, 5, 32, 4, 1)
    w0 = init_weights(cfg, 0)
    net = PolicyValueNet(cfg, weights=w0)                                  # the rank's self-play evaluator
    ptrs = {k: v.data_ptr() for k, v in net.w.items()}
    B = 16
    fill = [5 * B + 3, 2 * B + 9][rank]                                    # this rank's replay ring: rank 1 holds fewer batches
    rng = np.random.RandomState(100 + rank)
    states = torch.from_numpy((rng.rand(fill, 2, 7, 7) < 0.2).astype(np.float32))
    pis = torch.from_numpy(rng.dirichlet([0.3] * 49, size=fill).astype(np.float32))
    zs = torch.from_numpy(rng.choice([-1.0, 0.0, 1.0], size=(fill, 1)).astype(np.float32))
    steps = agree_min(fill // B, dist, "cpu")                              # main.py:35, agreed: every step carries one all-reduce
    tr = Trainer(cfg, w0, dropout=0.0)
    n_ar = [0]
    real_all_reduce = dist.all_reduce

    def counting(t, *a, **k):
        n_ar[0] += 1
        return real_all_reduce(t, *a, **k)
    dist.all_reduce = counting
    tr.train(((states[i * B:(i + 1) * B], pis[i * B:(i + 1) * B], zs[i * B:(i + 1) * B]) for i in range(steps)), 0.001, dist=dist)
    dist.all_reduce = real_all_reduce
    sd = tr.state_dict()
    in_place = net.load_state_dict(sd)                                     # promotion (main.py:59)
    same_addr = all(net.w[k].data_ptr() == p for k, p in ptrs.items())
    dig = hashlib.sha256(b"".join(sd[k].numpy().tobytes() for k in sorted(sd))).hexdigest()
    x = torch.zeros(1, 2, 7, 7)
    x[0, 0, 3, 3] = 1
    out = hashlib.sha256(net(x)[0].numpy().tobytes()).hexdigest()
    q.put((rank, steps, n_ar[0], in_place, same_addr, dig, out))
    dist.destroy_process_group()


def test_world2_train_and_in_place_promotion_with_unequal_replay_fills():
    """BASELINE config 5 readiness without a node: two gloo ranks whose replay rings hold 5 and 2 batches run
    the SAME number of train steps (one all-reduce each), end with identical weights (the data-parallel update), and promote
    them into their evaluators in place - identical outputs on both ranks, weight addresses unchanged."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29900 + (os.getpid() % 1500)
    procs = [ctx.Process(target=_train_promote_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=240) for _ in range(2))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (r0, s0, a0, ip0, sa0, d0, o0), (r1, s1, a1, ip1, sa1, d1, o1) = got
    assert (s0, s1) == (2, 2) and a0 == a1 == 2                   # the count is the minimum over the ranks, one bucket all-reduce per step
    assert ip0 and ip1 and sa0 and sa1
    assert d0 == d1 and o0 == o1                                   # same gradients everywhere -> same weights, same evaluator
