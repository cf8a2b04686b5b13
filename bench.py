#!/usr/bin/env python3
"""bench.py - self-play data generation throughput on MI355X (BASELINE.json metric).

  python bench.py [--gpus N] [--steps K] [--warmup W]            (N > 1: launched by torch.distributed.run)

Workload (BASELINE.json configs[2]): Gomoku 15x15, 800 simulations per move, 2048 concurrent self-play
games per GPU, policy-value net = the reference's training config (ai/nn.py Net(15, patch 5, embed 512,
225 actions, 8 heads, depth 1, 2 channels), random init, seed 0), synthetic start (empty boards).

A STEP is one move of the whole resident batch: 2048 searches x 800 simulations (tree kernels + leaf
compaction + network evaluation + expansion/backup), move selection, state advance, the per-move records
(board, pi, q, action) copied to the host, and finished games restarted (continuous self-play).
value = self-play games/sec = games COMPLETED inside the timed region / seconds.  All slots start in phase (empty
boards), and a completion count is only meaningful for a stationary process, so an UNTIMED pre-roll that does not depend
on --warmup de-phases the slots first: `--preroll-cheap` moves at `--preroll-sims` simulations (cheap: scrambles the
ages of the games) and then `--preroll-full` moves at the full simulation count (one mean game length: the games alive
when timing starts were played under the benchmark's own search).  After that the completion rate does not depend on
the window (--steps 20 --warmup 5 and --steps 40 --warmup 40 agree within noise) and the lengths of the games that
complete in the window are an unbiased sample, so the renewal estimate (plies / mean length / s) printed next to the
value agrees with it.

`python bench.py --gpus N` started WITHOUT torch.distributed.run spawns its N ranks itself (children, before any GPU
call); a world size that disagrees with --gpus is an error (non-zero exit), never a silent one-GPU run.

One JSON line on rank 0.  Extra objects:
  roofline      k_tree (PUCT scan + expand + backup), HBM-bound: algorithmic bytes per launch from the engine's
                device counters / mean launch duration from HIP events on the launch stream
  fp32_line     the same workload timed a second time IN THIS RUN on the fp32-accurate evaluator (the one that meets north_star's
                1e-5 bar on visit-count policies); `parity.quoted` holds builder-run numbers read from profiles/, labelled as such
  cpu_baseline  the oracle (CPU restatement of the reference algorithm, batch-1 fp32 net, eval cache) timed on
                this box's host cores on a bounded sample of the same workload
"""
import argparse
import json
import os
import sys
import time

T_PROCESS_START = time.perf_counter()
ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "alpha-zero_amd"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

MFMA_BF16_PEAK_TFLOPS = 2500.0      # dense bf16 / fp16 MFMA peak, MI355X_MICROARCH.md
MFMA_F32_PEAK_TFLOPS = 157.3        # f32-input MFMA (v_mfma_f32_16x16x4_f32) = the fp32 vector rate, MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
BF16_PEAK_TFLOPS = 2500.0      # dense bf16 MFMA


def survey_bytes(c, planes, rows, cols, action_dim):
    """SURVEY.md 8(d)'s algorithmic bytes of the tree kernels: 12 B per child scanned (N int32, W f32, P f32) + 16 B per path node
    (read-modify-write of N, W) + 12 B per child created + F R C 4 B (leaf board written for the network) and A 4 B (logits read)
    per evaluated leaf - the reference's fields at the survey's widths.  `roofline.frac` is priced on THIS formula."""
    return (12 * c["edges_scanned"] + 16 * c["trace_nodes"] + 12 * c["edges_created"]
            + c["leaves_evaluated"] * (planes * rows * cols * 4 + action_dim * 4))


def algorithmic_bytes(c, action_dim):
    """k_tree traffic model of THIS build's layout (`frac_layout_formula`: W is a float64 column, a header record is 16 B) (DESIGN.md 'k_tree roofline'): per scanned child N,W,P = 4+8+4 B; per backed-up
    node a read-modify-write of N and W = 2*(4+8) B; per created child N,W,P,cell,first_child,n_children =
    4+8+4+2+4+2 B plus its move-list entry written then read (2+2 B); per evaluated leaf the logits row (4*A B),
    the leaf board written (R*C B) and the path written then read (8 B per node, folded into the trace term)."""
    return (16 * c["edges_scanned"] + 24 * c["trace_nodes"] + 28 * c["edges_created"]
            + c["leaves_evaluated"] * (4 * action_dim + action_dim) + 8 * c["trace_nodes"])


# the reference itself (pure Python + batch-1 fp32 torch), timed at survey time in the build container:
# it cannot travel to the GPU box, so it is quoted, not re-measured
REFERENCE_PYTHON = {"games_per_sec": 0.0157, "sims_per_sec": 340.0, "seconds_per_game": 63.6,
                    "hardware": "8 vCPU Intel Xeon 2.10 GHz (build container), 8 torch intra-op threads",
                    "source": "BASELINE.md section 2: Gomoku 15x15, 800 sims/move, one 27-move game, seed 0; survey-time measurement, not this box"}


def cpu_port_sample(n_sims, budget_s, threads):
    """One bounded sample of the oracle (kind 'port'): sequential search in C, batch-1 float32 ViT through PyTorch CPU,
    eval cache on, numpy softmax - the reference's algorithm and evaluator shape.  Returns (sims, evals, hits, moves, seconds)."""
    import numpy as np
    import torch
    from oracle import az_oracle as ao
    from pvnet import NetConfig, PolicyValueNet
    torch.set_num_threads(threads)
    cfg = NetConfig(15, 15, 2, 225, 5, 512, 8, 1)
    net = PolicyValueNet(cfg, seed=0, device="cpu", dtype=torch.float32, path="full")
    game = ao.OracleGame("gomoku", 15)
    cache = ao.OracleCache(game)
    cnt = ao.Counters()
    rng = np.random.RandomState(0)

    def ev(canon):
        logits, v = net(torch.from_numpy(np.ascontiguousarray(canon))[None])
        l = logits[0].numpy()
        return np.exp(l) / np.sum(np.exp(l)), float(v[0, 0])
    t0 = time.time()
    # two half-samples, so that the searches timed are not all early-game ones (few children, almost no terminal leaves): a game
    # from the empty board, and one continued from ply 12 of a recorded 15x15 game of the reference (tests/golden/games.npz)
    out = ao.self_play(game, ev, n_sims, noise_fn=lambda mc: rng.dirichlet([0.03] * 225),
                       uniform_fn=lambda mc: rng.random_sample(), cache=cache, counters=cnt, time_budget=budget_s * 0.5)
    moves = len(out["cells"])
    try:
        from conftest import golden_meta, load_golden
        zg = load_golden("games.npz")
        m15 = next(m for m in golden_meta(zg) if m["size"] == 15 and len(zg[f"g{m['game']}_board_cells"]) > 14)
        cells = zg[f"g{m15['game']}_board_cells"][12].reshape(15, 15)
        board = np.stack([(cells == 1), (cells == 2)]).astype(np.float32)
        out2 = ao.self_play(game, ev, n_sims, noise_fn=lambda mc: rng.dirichlet([0.03] * 225), uniform_fn=lambda mc: rng.random_sample(),
                            cache=cache, counters=cnt, time_budget=budget_s * 0.5, start=(board, 0, 12))
        moves += len(out2["cells"])
    except StopIteration:
        pass
    return cnt.mcts_count, cnt.evals, cnt.matched, moves, time.time() - t0


def cpu_baseline(n_sims, budget_s, mean_plies):
    """cpu_baseline object of the JSON line.  Two legs on this box's host cores, each a bounded sample of the same workload:
    (1) one process, 8 intra-op threads (batch-1 evaluation does not scale past a few threads; 8 is what the survey's
    reference measurement used) - the headline `value`; (2) one single-threaded process per core of this box's CPU share
    (16 per GPU), the 'all host cores' figure of SURVEY 8(d).  games/s = sims/s / (sims per move x mean plies per game),
    with the same mean game length the GPU line reports."""
    import subprocess
    import torch
    threads = min(8, os.cpu_count() or 1)
    sims, evals, hits, moves, dt = cpu_port_sample(n_sims, budget_s * 0.5, threads)
    sims_per_s = sims / dt
    games_per_s = sims_per_s / (n_sims * mean_plies)
    out = {"value": games_per_s, "unit": "games/s", "cores": threads, "kind": "port",
           "sims_per_sec": sims_per_s, "host_cpus": os.cpu_count(),
           "sample": f"{moves} searches of Gomoku 15x15 at {n_sims} sims/move - half of the time from the empty board, half continued from ply 12 of a recorded game: {sims} sims "
                     f"({evals} net evals, {hits} cache hits) in {dt:.1f} s; games/s = sims/s / ({n_sims} x {mean_plies:.1f} plies)",
           "reference_python": REFERENCE_PYTHON}
    # leg 2: independent single-threaded processes, one per core of the CPU share; children never touch the GPU
    nproc = min(16, os.cpu_count() or 1)
    env = dict(os.environ, HIP_VISIBLE_DEVICES="-1", ROCR_VISIBLE_DEVICES="-1", OMP_NUM_THREADS="1", MKL_NUM_THREADS="1")
    cmd = [sys.executable, os.path.abspath(__file__), "--cpu-worker", str(budget_s * 0.5), "--sims", str(n_sims)]
    t0 = time.time()
    procs = [subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True) for _ in range(nproc)]
    tot, ok = 0.0, 0
    for pr in procs:
        try:
            so, _ = pr.communicate(timeout=budget_s * 0.5 + 240)
            r = json.loads(so.strip().splitlines()[-1])
            tot += r["sims"] / r["seconds"]
            ok += 1
        except Exception:
            pr.kill()
    out["all_cores"] = {"processes": ok, "threads_per_process": 1, "sims_per_sec": tot, "games_per_sec": tot / (n_sims * mean_plies),
                        "wall_s": time.time() - t0,
                        "note": "independent games, one single-threaded process per core of this box's CPU share (16 per GPU); summed"}
    return out


class StubRunner:
    """Engine-shaped stand-in (launcher / reduction tests on hosts without a GPU, `--stub-engine`): the SelfPlayRunner
    surface bench.py uses, with deterministic per-rank work.  Its JSON line says data = "stub"; it measures nothing."""

    def __init__(self, games, sims, first_global_game):
        self.G, self.n_sims, self.first, self.n_split = games, sims, first_global_game, 1
        self.plies_played = self.games_finished = self.finished_plies = self.move_idx = 0
        self._c = {"sims": 0, "edges_scanned": 0, "trace_nodes": 0, "edges_created": 0, "leaves_evaluated": 0, "cache_hits": 0}

    def play_move(self):
        self.move_idx += 1
        self.plies_played += self.G
        fin = (self.first + self.move_idx) % 7 + 1            # depends on the shard: ranks report different work
        self.games_finished += fin
        self.finished_plies += 20 * fin
        self._c["sims"] += self.G * self.n_sims
        self._c["leaves_evaluated"] += self.G * self.n_sims // 2

    def counters(self):
        return dict(self._c)

    def reset_counters(self):
        self._c = {k: 0 for k in self._c}

    def check_error(self):
        pass


def spawn_ranks(n, argv):
    """`python bench.py --gpus N` outside torch.distributed.run: start the N ranks as a child job (the driver's own
    command line) before this process has touched the GPU, and exit with its code."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "8")
    return subprocess.call(cmd, env=env)


def fp32_window(args, cfg, rank, local_rank, world, dist):
    """The second timed window of the default run: the SAME workload on the hand-written fp32-accurate evaluator (csrc/azk_nnx.hip +
    k_embed_fold<EX>) - the evaluator whose 800-simulation visit-count policies equal the reference's float32 network's on every
    recorded position (tests/test_gpu_exact.py, asserted at north_star's 1e-5) - with its own untimed pre-roll, timed like the
    headline (barrier + synchronize on both sides, MAX over ranks).  Returns the object printed as `fp32_line`."""
    import torch
    from pvnet import PolicyValueNet
    from selfplay import SelfPlayRunner
    from shard import reduce_measurement, shard_range
    net = PolicyValueNet(cfg, seed=0, device=f"cuda:{local_rank}", dtype=torch.float32, path="clsfold")
    hand_written = getattr(net, "_exact", None) is not None
    net.use_chain_tail, net.use_fold_u = True, True
    runner = SelfPlayRunner("gomoku", net, args.games, args.sims, size=args.size, seed=args.seed,
                            first_global_game=shard_range(args.games, rank)[0], device=local_rank, leaf_dtype="float32",
                            recycle=True, kernel_timer=None, use_graph=True, n_split=1, cache_entries=args.cache_entries,
                            cache_shared=args.cache == "shared", steps_per_graph=args.steps_per_graph)

    def sync_all():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()
    if args.preroll_cheap > 0:
        runner.n_sims = min(args.sims, max(8, args.preroll_sims))
        for _ in range(args.preroll_cheap):
            runner.play_move()
        runner.n_sims = args.sims
    for _ in range(args.preroll_full + args.fp32_warmup):
        runner.play_move()
    runner.reset_counters()
    plies0, fin0, finp0 = runner.plies_played, runner.games_finished, runner.finished_plies
    sync_all()
    t0 = time.perf_counter()
    for _ in range(args.fp32_steps):
        runner.play_move()
    sync_all()
    dt = time.perf_counter() - t0
    runner.check_error()
    net.check_exact_range()
    c = runner.counters()
    dt_max, (plies_all, fin_w, finp_w, sims_all, leaves_all) = reduce_measurement(
        dt, [runner.plies_played - plies0, runner.games_finished - fin0, runner.finished_plies - finp0, c["sims"], c["leaves_evaluated"]],
        dist if world > 1 else None, "cuda")
    for h in runner.halves:
        h.eng.close()
    return {"games_per_sec": fin_w / dt_max, "sims_per_sec": sims_all / dt_max, "ms_per_step": dt_max / args.fp32_steps * 1e3,
            "steps": args.fp32_steps, "warmup": args.fp32_warmup, "seconds": dt_max, "games_finished_in_window": fin_w,
            "games_per_sec_renewal_estimate": (plies_all / (finp_w / fin_w) / dt_max) if fin_w else None,
            "leaf_evals_per_sec": leaves_all / dt_max, "dtype": "f32", "measured": "in this run, after the headline window",
            "evaluator": ("hand-written fp32-accurate kernels (k_embed_fold<EX> + k_gemm_h x 5, csrc/azk_nnx.hip)" if hand_written
                          else "torch float32 library forward (the hand-written fp32 path does not cover this network)"),
            "parity": "visit-count policies identical to the reference's float32 network on all recorded positions (tests/test_gpu_exact.py, asserted at 1e-5)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--preroll-cheap", type=int, default=128, help="untimed de-phasing moves at --preroll-sims simulations")
    ap.add_argument("--preroll-sims", type=int, default=16)
    ap.add_argument("--preroll-full", type=int, default=26, help="untimed moves at the full simulation count after the cheap pre-roll")
    ap.add_argument("--stub-engine", action="store_true", help="TEST ONLY: gloo on the CPU with a stub runner (launcher / reduction tests)")
    ap.add_argument("--cpu-worker", type=float, default=0.0, help="internal: one single-threaded CPU-baseline sample of this many seconds")
    ap.add_argument("--games", type=int, default=2048, help="concurrent games per GPU")
    ap.add_argument("--sims", type=int, default=800)
    ap.add_argument("--size", type=int, default=15)
    ap.add_argument("--nn-path", default=None, choices=["clsfold", "cls", "full"],
                    help="default: clsfold (hand-written kernels: azk_nn.hip for bf16, the fp32-accurate azk_nnx.hip for fp32); cls / full = torch library forward")
    ap.add_argument("--nn-dtype", default="bf16", choices=["bf16", "fp32"],
                    help="bf16 (default, north_star's MFMA bf16): the hand-written bf16 kernels; fp32: the hand-written fp32-accurate kernels "
                         "(exact products of the 0/1 board with fp16 hi/lo terms, float32 elsewhere) - the evaluator whose visit-count policies "
                         "equal the reference's float32 network's to the last visit")
    ap.add_argument("--tail", default="chain", choices=["chain", "library"],
                    help="cls-row tail: the all-hand-written GEMM chain (azk_nn_tail_gemm: honours the live leaf count; default) or "
                         "hipBLASLt GEMMs + the hand-written LN / heads kernels (always runs the full 2048-row buffer)")
    ap.add_argument("--no-graph", action="store_true", help="eager stepping with a host sync per simulation (n_leaf-sized batches)")
    ap.add_argument("--embed", choices=("fold", "conv"), default="fold", help="bf16 embedding + pooling: fold = k_embed_fold (patch-pooling form), conv = k_embed_pool_c")
    ap.add_argument("--split", type=int, default=1, help="independent game groups stepped on separate streams inside the step graph")
    ap.add_argument("--cache-entries", type=int, default=32768, help="per-game eval-cache entries (MCTS.cache; 64 GB of HBM at 2048 games x 32768); 0 = off")
    ap.add_argument("--cache", default="shared", choices=["shared", "per-game"],
                    help="eval cache: one table shared by every game of the GPU (the reference's MCTS.cache is process-global, mcts.py:7) or one "
                         "table per game; same memory, same results, different hit rate")
    ap.add_argument("--timer-stride", type=int, default=176,
                    help="every n-th simulation step runs eagerly with HIP events around k_tree and the embedding kernel (the roofline's live durations)")
    ap.add_argument("--steps-per-graph", type=int, default=32,
                    help="simulation steps captured in one hipGraph (consecutive graph launches leave an ~8 us bubble; 1 = one step per launch)")
    ap.add_argument("--budget-stepping", type=int, default=0,
                    help="1: a game keeps simulating inside a tree launch while its simulations need no evaluator (terminal leaves, eval-cache "
                         "hits) - same trees.  Measured SLOWER under per-move lock-step (236 vs 126 ms/move): the launch count of a move is set "
                         "by its slowest game (early-ply games miss the cache almost always) while every launch lasts as long as its busiest "
                         "wave; it needs asynchronous moves to pay (DESIGN.md section 10).  0 (default): one simulation per game and launch")
    ap.add_argument("--async-moves", type=int, default=0,
                    help="1: asynchronous per-game moves (a game moves as soon as ITS search is done: azk_async_*) with budget stepping - the same "
                         "trees, moves and games as lock-step (tests/test_gpu_async.py); a step is then G moves of the batch in total")
    ap.add_argument("--per-launch", type=int, default=2, help="with --async-moves: most simulations a game runs inside one tree launch")
    ap.add_argument("--young-us", type=int, default=0, help="with --async-moves: a game starts another simulation inside a launch only while the launch is younger than this many microseconds (0: no limit)")
    ap.add_argument("--virtual-loss", type=int, default=1, metavar="K",
                    help="OPT-IN, NOT the headline: K > 1 leaves in flight per game with a virtual loss on their paths (north_star's 'virtual-loss "
                         "expansion').  Changes search results (the reference's search is sequential), so the line is reported under its own metric key")
    ap.add_argument("--cpu-seconds", type=float, default=20.0, help="wall-clock budget of the CPU baseline leg (0 = skip)")
    ap.add_argument("--fp32-steps", type=int, default=12,
                    help="the default (bf16) run ends with a second timed window on the fp32-accurate evaluator (`fp32_line`): this many moves (0 = skip)")
    ap.add_argument("--fp32-warmup", type=int, default=4, help="untimed moves of the fp32 window after its own pre-roll")
    ap.add_argument("--net", default="main", choices=["main", "compare"],
                    help="main (default): main.py:134's network, Net(patch 5, embed 512, heads 8, depth 1) - the metric's config.  compare: SIDE LINE on "
                         "main.py:186-188's network, Net(patch 5, embed 256, heads 8, depth 2): every block on the hand-written full-token kernels "
                         "(csrc/azk_block.hip), evaluated over the fixed-size leaf buffer with the device-side live count")
    ap.add_argument("--tail-wide", default="lds", choices=["lds", "registers"],
                    help="the two wide links of the chain tail: LDS-staged (csrc/azk_tail.hip, default) or round 3's whole-K-in-registers form (same-box A/B)")
    ap.add_argument("--split-fit", type=int, default=0,
                    help="with --split N: shape the network kernels so that another group's tree waves fit beside them - k_embed_fold launches at most this "
                         "many workgroups (256 = one per CU), the K = 2048 tail link keeps two LDS ring buffers")
    ap.add_argument("--blocks", default="hip", choices=["hip", "library"],
                    help="--net compare: the full-token blocks on the hand-written kernels (default) or on the torch library (F.linear / SDPA)")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--train-step", action="store_true",
                    help="BASELINE.json configs[4]: one train.py step (batch 512 per GPU, Adam lr 2.5e-4, one fused gradient bucket "
                         "all-reduced over RCCL) after every move, fed from the device-resident replay ring")
    ap.add_argument("--promote-every", type=int, default=8,
                    help="with --train-step: every n-th move the trained weights are promoted into the self-play evaluator (main.py:55-59: "
                         "load_state_dict + MCTS.cache.clear()), in place - the captured step graphs keep replaying; 0 = never")
    args = ap.parse_args()
    if args.nn_path is None:
        args.nn_path = "clsfold"

    if args.cpu_worker > 0:              # child of cpu_baseline's all-cores leg: CPU only, prints one JSON line
        sims, evals, hits, moves, dt = cpu_port_sample(args.sims, args.cpu_worker, 1)
        print(json.dumps({"sims": sims, "evals": evals, "hits": hits, "moves": moves, "seconds": dt}))
        return 0

    from shard import agree_min, env_world, shard_range
    rank, local_rank, world = env_world()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        return spawn_ranks(args.gpus, sys.argv[1:])             # children; this process never touches the GPU
    if world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: refusing to report a {world}-rank run as {args.gpus} GPUs", file=sys.stderr)
        return 2

    import torch
    import torch.distributed as dist
    stub = args.stub_engine
    if not stub and torch.cuda.device_count() < max(1, local_rank + 1):
        print(f"bench.py: rank {rank} needs cuda:{local_rank} but {torch.cuda.device_count()} GPU(s) are visible", file=sys.stderr)
        return 3
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if stub:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    torch.set_num_threads(min(8, os.cpu_count() or 1))       # the host only launches kernels; N ranks must not each spawn 128 threads
    rdev = "cpu" if stub else "cuda"

    from pvnet import NetConfig
    A = args.size * args.size
    cfg = NetConfig(args.size, args.size, 2, A, 5, 512, 8, 1) if args.net == "main" else NetConfig(args.size, args.size, 2, A, 5, 256, 8, 2)
    if args.net != "main":
        args.fp32_steps = 0
    replay = trainer = None
    train_ms = []
    if stub:
        class _NoTimer:
            enabled, pairs, children = False, [], {}

            def mean_ms(self):
                return None
        kt, net = _NoTimer(), None
        runner = StubRunner(args.games, args.sims, shard_range(args.games, rank)[0])
    else:
        torch.cuda.set_device(local_rank)
        from pvnet import PolicyValueNet
        from selfplay import KernelTimer, SelfPlayRunner
        nn_torch_dtype = torch.bfloat16 if args.nn_dtype == "bf16" else torch.float32
        net = PolicyValueNet(cfg, seed=0, device=f"cuda:{local_rank}", dtype=nn_torch_dtype, path=args.nn_path)
        net.use_chain_tail = args.tail == "chain"
        net.use_hip_blocks = args.blocks == "hip"
        net.use_lds_tail = args.tail_wide == "lds"
        if args.split > 1 and args.split_fit:
            import azk
            azk.lib().azk_nn_tail_lds_footprint(1)
            azk.lib().azk_nn_embed_fold_grid(args.split_fit)
        kt = KernelTimer(stride=args.timer_stride)
        exact = getattr(net, "_exact", None) is not None and args.nn_path == "clsfold"      # fp32: the hand-written fp32-accurate kernels (csrc/azk_nnx.hip)
        net.use_fold_u = args.embed == "fold"
        fold = (not exact and getattr(net, "_foldu", None) is not None and net.use_fold_u and args.tail == "chain" and getattr(net, "chain_tail", False)
                and args.nn_path == "clsfold")         # k_embed_fold: patch-pooling form, token rows never formed
        xfold = (exact and net._exact.get("foldu") is not None and net.use_fold_u and getattr(net, "exact_tail", "f32") == "h16")   # k_embed_fold<EX>
        fold = fold or xfold
        ep_tables = (net._exact["foldu"] if xfold else net._exact["tables"]) if exact else (net._foldu if fold else getattr(net, "_compact", None))
        ep_stats = ep_tables.enable_work_stats() if ep_tables is not None else None   # device counters: boards / 16-token tiles evaluated
        if args.train_step:
            from azk import DeviceReplay
            from trainer import Trainer
            replay = DeviceReplay(400000, cfg.channels, cfg.rows, cfg.cols, cfg.action_dim, device=torch.device("cuda", local_rank))
            trainer = Trainer(cfg, net.state_dict(), device=f"cuda:{local_rank}", dropout=0.1)      # main.py:134 trains with dropout 0.1
        if args.async_moves:
            from selfplay import AsyncSelfPlayRunner
            runner = AsyncSelfPlayRunner("gomoku", net, args.games, args.sims, size=args.size, seed=args.seed,
                                         first_global_game=shard_range(args.games, rank)[0], device=local_rank,
                                         leaf_dtype="bfloat16" if args.nn_dtype == "bf16" else "float32", recycle=True, kernel_timer=kt,
                                         cache_entries=args.cache_entries, cache_shared=args.cache == "shared", replay=replay,
                                         per_launch=args.per_launch, steps_per_graph=args.steps_per_graph, use_graph=not args.no_graph,
                                         young_launch_us=args.young_us)
        else:
            runner = SelfPlayRunner("gomoku", net, args.games, args.sims, size=args.size, seed=args.seed,
                                    first_global_game=shard_range(args.games, rank)[0], device=local_rank,
                                    leaf_dtype="bfloat16" if args.nn_dtype == "bf16" else "float32",
                                    recycle=True, kernel_timer=kt, use_graph=not args.no_graph, n_split=args.split, cache_entries=args.cache_entries,
                                    cache_shared=args.cache == "shared", replay=replay, budget_stepping=bool(args.budget_stepping),
                                    steps_per_graph=args.steps_per_graph, leaves_per_step=args.virtual_loss)

    def train_one():
        """train.train with one iteration (train.py:85-123): fresh Adam, the reference's loss, gradient bucket all-reduce.
        Whether a step runs is agreed by ALL ranks (MIN of 'my ring holds a batch'): the step carries an all-reduce, and a
        rank's replay fill depends on its own game lengths."""
        if not agree_min(1 if replay.size() >= 512 else 0, dist if world > 1 else None, rdev):
            return
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        trainer.train([replay.sample(512)], 0.00025, dist=dist if world > 1 else None)
        b.record()
        train_ms.append((a, b))

    promote_ms = []

    def promote():
        """main.py:55-59: the new weights become the self-play model and MCTS.cache is cleared.  The evaluator's device buffers are
        refreshed in place (PolicyValueNet.load_state_dict), so the captured step graphs stay valid."""
        torch.cuda.synchronize()
        t_ = time.perf_counter()
        in_place = net.load_state_dict(trainer.state_dict())
        if not in_place:
            runner._graph = None
        for h in runner.halves:
            h.eng.clear_cache()
        torch.cuda.synchronize()
        promote_ms.append(((time.perf_counter() - t_) * 1e3, in_place))

    def sync_all():
        if not stub:
            torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            if not stub:
                torch.cuda.synchronize()

    # ---- untimed pre-roll (always, whatever --warmup says): de-phase the slots, then one game length under the real search ----
    if args.preroll_cheap > 0:
        full = runner.n_sims
        runner.n_sims = min(full, max(8, args.preroll_sims))
        for _ in range(args.preroll_cheap):
            runner.play_move()
        runner.n_sims = full
    for _ in range(args.preroll_full):
        runner.play_move()
    for _ in range(args.warmup):
        runner.play_move()
        if args.train_step:
            train_one()
    train_ms.clear()
    if hasattr(runner, "finish"):
        runner.finish()             # asynchronous runner: its host-side statistics lag the device by up to two chunks - bring them up to date
    runner.reset_counters()         # BEFORE the start-of-window snapshot, or the window is credited moves enqueued during the warm-up
    if not stub and ep_stats is not None:
        ep_stats.zero_()
    plies0, fin0, finp0 = runner.plies_played, runner.games_finished, runner.finished_plies
    launches0 = getattr(runner, "launches", 0)
    if os.environ.get("AZK_DUMP_MAPS"):     # diagnosis of a profiler crash: the process's mappings, to turn a stack trace's addresses into library offsets
        open(os.environ["AZK_DUMP_MAPS"], "w").write(open("/proc/self/maps").read())
    kt.enabled = True
    sync_all()
    t0 = time.perf_counter()
    for i_ in range(args.steps):
        runner.play_move()
        if args.train_step:
            train_one()
            if args.promote_every > 0 and (i_ + 1) % args.promote_every == 0 and train_ms:
                promote()
    sync_all()
    dt = time.perf_counter() - t0
    if hasattr(runner, "finish"):
        runner.finish()                                     # asynchronous runner: the final statistics of everything it enqueued
    kt.enabled = False
    runner.check_error()
    c = runner.counters()
    plies = runner.plies_played - plies0
    fin, finp = runner.games_finished, runner.finished_plies

    # max time / summed work over ranks (the only collectives: measurement, never the generation path)
    from shard import reduce_measurement
    dt_max, (plies_all, fin_window, finp_window, sims_all, leaves_all) = reduce_measurement(
        dt, [plies, fin - fin0, finp - finp0, c["sims"], c["leaves_evaluated"]], dist if world > 1 else None, rdev)

    # ---- second timed window of the default run: the same workload on the fp32-accurate evaluator (north_star's 1e-5 parity bar) ----
    fp32 = None
    if (not stub and args.nn_dtype == "bf16" and args.fp32_steps > 0 and args.nn_path == "clsfold" and not args.async_moves
            and args.virtual_loss <= 1 and not args.train_step and args.split == 1 and not args.no_graph and not args.budget_stepping):
        n_split_, spg_, lps_ = runner.n_split, runner.steps_per_graph, runner.leaves_per_step
        for h in runner.halves:
            h.eng.close()                               # the headline engine's arena and cache (~75 GB) go back before the second one is built
        runner._graph = runner._graph_many = None
        torch.cuda.empty_cache()
        fp32 = fp32_window(args, cfg, rank, local_rank, world, dist)

    if rank == 0:
        # after the pre-roll the process is stationary: the games that COMPLETE in the window are an unbiased sample of the
        # game-length distribution (it is the games ALIVE at an instant that are length-biased, not the ones ending in a window)
        if fin_window > 0:
            mean_plies, src = finp_window / fin_window, f"{int(fin_window)} games completed inside the timed window"
        else:
            mean_plies, src = 26.0, "no game completed in the window: assumed 26 plies"
        est_games_per_s = plies_all / mean_plies / dt_max          # renewal estimate: plies per second / mean game length
        games_per_s, value_src = fin_window / dt_max, "games completed inside the timed window / seconds (slots de-phased by the untimed pre-roll)"
        if stub:
            print(json.dumps({"metric": "selfplay_games_per_sec", "value": games_per_s, "unit": "games/s", "n_gpus": world, "steps": args.steps,
                              "warmup": args.warmup, "ms_per_step": dt_max / args.steps * 1e3, "data": "stub", "sims_per_sec": sims_all / dt_max,
                              "games_finished_in_window": fin_window, "plies_in_window": plies_all}))
            if world > 1:
                dist.destroy_process_group()
            return 0
        ev_dropped = [0, 0]                       # HIP-event samples dropped as host stalls / taken (KernelTimer.robust_mean_ms)

        def rmean(t_):
            m_, d_, n_ = t_.robust_mean_ms()
            ev_dropped[0] += d_; ev_dropped[1] += n_
            return m_
        tree_ms = rmean(kt)
        launches = max(1, (getattr(runner, "launches", 0) - launches0)) * runner.n_split      # k_tree launches in the window (one per game group and step)
        alg_bytes = algorithmic_bytes(c, A) / launches                         # this build's layout
        sv_bytes = survey_bytes(c, cfg.channels, cfg.rows, cfg.cols, A) / launches  # SURVEY 8(d)'s formula: what `frac` is priced on
        roof = None
        traffic, traffic_src, pmc = None, None, {}
        try:        # HBM bytes per k_tree launch from the PMC passes committed under profiles/ (FETCH_SIZE + WRITE_SIZE): QUOTED, not measured in this run
            pmc_file = next(f for f in ("r04_pmc_traffic.json", "r03_pmc_traffic.json") if os.path.exists(os.path.join(ROOT, "profiles", f)))
            pmc = json.load(open(os.path.join(ROOT, "profiles", pmc_file)))
            traffic = next(v["bytes_per_launch"] for k_, v in pmc.items() if k_.startswith("k_tree<true, true"))
            traffic_src = f"QUOTED from profiles/{pmc_file} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes on the same workload; a counter pass cannot run inside this process)"
        except Exception:
            pass
        # The HIP-event pairs bracket kernels of steps that run EAGERLY inside the timed window (every --timer-stride-th step): each pair
        # also times its own event records and the gap of an eager launch, so the raw samples over-read the durations the kernels have
        # inside the captured graphs (their sum exceeds the measured step).  The step time itself IS measured (window / launches), and the
        # over-read is per PAIR, so every sample is reduced by (sum of the samples - measured step time) / (number of pairs per step):
        # `avg_launch_us` is that (it reproduces the rocprofv3 averages within ~0.3 us), the raw sample is kept beside it.
        ev_us = {"tree": (tree_ms or 0.0) * 1e3}
        for nm_ in ("k_embed", "k_cls_pool", "k_embed_pool"):
            ch_ = kt.children.get(nm_)
            ev_us[nm_] = (rmean(ch_) or 0.0) * 1e3 if ch_ else 0.0
        ch_ = kt.children.get("k_tail")
        tail_launches = 5 if (args.nn_path == "clsfold" and (args.nn_dtype == "fp32" or args.tail == "chain")) else 1
        ev_us["k_tail"] = (rmean(ch_) or 0.0) * 1e3 * tail_launches if ch_ else 0.0
        step_us = dt * 1e6 / max(1, launches / runner.n_split)
        ev_sum = sum(ev_us.values())
        ev_pairs = {k_: (tail_launches if k_ == "k_tail" else 1) for k_, v_ in ev_us.items() if v_ > 0}
        ev_over = max(0.0, (ev_sum - step_us) / max(1, sum(ev_pairs.values()))) if (ev_sum > 0 and runner.n_split == 1 and not args.no_graph) else 0.0

        def ev_fix(ms_, pairs_=1):                  # raw HIP-event mean (ms) -> duration inside the graph (ms)
            return max(ms_ * 0.5, ms_ - pairs_ * ev_over * 1e-3)
        dur_note = (f"HIP-event samples of eagerly launched steps inside the timed window, each reduced by {ev_over:.2f} us per event pair = (sum of the raw samples "
                    f"{ev_sum:.1f} us - measured step time {step_us:.1f} us [window / {launches // runner.n_split} simulation steps]) / {sum(ev_pairs.values())} pairs per step; "
                    f"{ev_dropped[0]} of {ev_dropped[1]} samples above 3x their kernel's median (host stalls between eager launches) dropped; "
                    "rocprofv3 --kernel-trace --stats of the same command is under profiles/")
        if tree_ms:
            t_us = ev_fix(tree_ms) * 1e3
            gbs = sv_bytes / (t_us * 1e-6) / 1e9
            roof = {"kernel": "k_tree<expand,select>", "bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": gbs / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src, "avg_launch_us": t_us,
                    "avg_launch_us_eager_sample": tree_ms * 1e3, "duration_source": dur_note,
                    "algorithmic_bytes_per_launch": sv_bytes,
                    "algorithmic_bytes_formula": "SURVEY.md 8(d): 12 B x children scanned + 16 B x path nodes + 12 B x children created + (F R C 4 + A 4) B x leaves evaluated, from the engine's device counters",
                    "layout_bytes_per_launch": alg_bytes, "frac_layout_formula": alg_bytes / (t_us * 1e-6) / 1e9 / HBM_PEAK_GBS,
                    "layout_formula": "this build's record widths: 16 B per child scanned (16-B header) + 32 B per path node (N, float64 W, path entry) + 28 B per child created + 5 A B per leaf",
                    "event_samples": len(kt.pairs)}
        from pvnet import flops_clsfold
        # per-kernel rooflines (HBM-bound streaming kernels): algorithmic bytes per launch / HIP-event duration
        live = leaves_all / max(1.0, launches * world / runner.n_split)       # boards the network kernels really process per launch
        T_tok, Dm = cfg.tokens, cfg.embed_dim
        kernels = []
        if roof:
            kernels.append(dict(roof))
        for name, per_board in (("k_embed", T_tok * Dm * 2 + cfg.num_heads * 4 * ((T_tok + 15) // 16 * 16) + 2 * cfg.rows * cfg.cols * 2),
                                ("k_cls_pool", T_tok * Dm * 2 + cfg.num_heads * Dm * 2 + cfg.num_heads * 4 * ((T_tok + 15) // 16 * 16))):
            ch = kt.children.get(name)
            ms = ch.robust_mean_ms()[0] if ch else None
            if ms:
                by = per_board * live
                kernels.append({"kernel": name, "bound": "hbm", "achieved": by / (ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                "frac": by / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "avg_launch_us": ms * 1e3, "duration_source": "raw HIP-event sample (eager step)", "algorithmic_bytes_per_launch": by,
                                "traffic": None, "event_samples": len(ch.pairs),
                                "note": f"{per_board} B per live board x {live:.0f} live boards per launch (device-side count)"})
        ch = kt.children.get("k_embed_pool")            # fused embedding + cls pooling: on-chip, priced against the dense bf16 MFMA peak
        ms = ch.robust_mean_ms()[0] if ch else None
        if ms:
            kreal = cfg.channels * cfg.patch_size ** 2
            per_board = 2 * (T_tok - 1) * Dm * kreal + 2 * T_tok * 16 * kreal + 2 * T_tok * cfg.num_heads * Dm
            fl = per_board * live
            ep_traffic = None
            try:
                # the counters were collected on the bf16 build's kernel: no figure for k_embed_pool_x unless the file holds one
                ep_traffic = next(v["bytes_per_launch"] for k_, v in pmc.items() if k_ == ("k_embed_fold" if fold and not exact else ("k_embed_pool_x" if exact and not fold else ("k_embed_pool_c" if not exact else "-"))))
            except Exception:
                pass
            compact = exact or fold or (getattr(net, "_compact", None) is not None and getattr(net, "use_compact", False))
            executed_share = None
            if compact and ep_stats is not None:
                # share of the 16-token tiles the compacting kernel really evaluated IN THIS RUN (device counters of the kernel itself)
                eb, et = (int(v) for v in ep_stats.tolist())
                executed_share = et / max(1, eb) / ((T_tok + 15) // 16)
                if fold:
                    # k_embed_fold issues 24 MFMAs of 16 x 16 x 32 per evaluated tile (quadratic form 16, score columns 4, pooled patch 4)
                    executed_share = et * 24 * 2 * 16 * 16 * 32 / max(1, eb) / per_board
            ep_peak = MFMA_BF16_PEAK_TFLOPS
            if exact and not fold:
                # two matrix pipes rates in one kernel: conv / score columns on the fp16 pipe (weights as hi + lo: two MFMAs per product),
                # the weighted token sum on v_mfma_f32_16x16x4_f32.  peak = the blended rate at which the algorithmic flops could issue
                pool_fl = 2 * T_tok * cfg.num_heads * Dm
                ep_peak = per_board / ((per_board - pool_fl) / MFMA_BF16_PEAK_TFLOPS + pool_fl / MFMA_F32_PEAK_TFLOPS)
            ms_raw, ms = ms, ev_fix(ms)
            kernels.append({"kernel": ("k_embed_fold<EX>" if fold else "k_embed_pool_x") if exact else ("k_embed_fold" if fold else ("k_embed_pool_c" if compact else "k_embed_pool")), "bound": "mfma", "achieved": fl / (ms * 1e-3) / 1e12,
                            "peak": ep_peak, "unit": "TFLOP/s", "frac": fl / (ms * 1e-3) / 1e12 / ep_peak,
                            "frac_issued": (fl * executed_share / (ms * 1e-3) / 1e12 / ep_peak) if (compact and executed_share) else None,
                            "avg_launch_us": ms * 1e3, "avg_launch_us_eager_sample": ms_raw * 1e3, "duration_source": dur_note,
                            "median_launch_us": ch.spread_us()[0], "max_launch_us": ch.spread_us()[1],
                            "algorithmic_flops_per_launch": fl, "traffic": ep_traffic,
                            "traffic_source": traffic_src if ep_traffic else None, "event_samples": len(ch.pairs),
                            "executed_share_of_algorithmic_flops": executed_share if compact else 1.0,
                            "note": (f"{per_board} flop per live board (the function's conv + score columns + weighted token sum over all {T_tok} tokens) x "
                                     f"{live:.0f} live boards per launch; ") +
                                    ("k_embed_fold computes the same function from the patch bits - LayerNorm's variance as a quadratic form, scores linear, "
                                     "the pooled row as token weights + pooled patch for the next GEMM - and only for the tokens a stone can reach: "
                                     "executed_share_of_algorithmic_flops = MFMA flops it issues / the function's; ~6 KB out per board" if fold else
                                     "the compacting kernel evaluates only the tokens a stone can reach and takes the "
                                     "rest as precomputed constants (executed_share_of_algorithmic_flops: MFMA work actually issued); HBM traffic is "
                                     "~9 KB per board (board in, z out): bound by VALU/MFMA issue and LDS, not HBM")})
        ch = kt.children.get("k_tail")                  # the cls-row tail (five k_tail_gemm launches, or the library GEMMs): MFMA-bound
        ms = ch.robust_mean_ms()[0] if ch else None
        ms_raw = None
        if ms and args.nn_path == "clsfold" and (exact or args.tail == "chain"):
            ms *= 5                                     # the timer holds one event pair per launch of the five-launch chain: their sum per step
        if ms:
            ms_raw, ms = ms, ev_fix(ms, tail_launches)
        if ms and args.nn_path == "clsfold":
            Hh, dh = cfg.num_heads, Dm // cfg.num_heads
            # value projection per head + output projection + MLP up + MLP down + merged heads, per row
            per_row = 2 * Hh * dh * Dm + 2 * Dm * Dm + 2 * 2 * Dm * 4 * Dm + 2 * Dm * (cfg.action_dim + 1)
            rows = live if (args.tail == "chain" or exact) else args.games * runner.leaves_per_step
            fl = per_row * rows
            h16 = exact and getattr(net, "exact_tail", "f32") == "h16"
            # the fp32-accurate tail: on fp16 (hi, lo) operand planes every product is three fp16 MFMAs (peak = a third of the fp16 rate),
            # on the float32-input MFMA it runs at the float32 vector rate
            tail_peak = (MFMA_BF16_PEAK_TFLOPS / 3 if h16 else MFMA_F32_PEAK_TFLOPS) if exact else MFMA_BF16_PEAK_TFLOPS
            kernels.append({"kernel": ("k_gemm_h x5 (cls-row tail, fp16 hi/lo planes)" if h16 else "k_gemm_x x5 (cls-row tail, f32 MFMA)") if exact
                            else ("k_tail_gemm x3 + k_tail_lds x2 (cls-row tail)" if args.tail == "chain" and getattr(net, "use_lds_tail", False) else ("k_tail_gemm x5 (cls-row tail)" if args.tail == "chain" else "library tail")), "bound": "mfma",
                            "achieved": fl / (ms * 1e-3) / 1e12, "peak": tail_peak, "unit": "TFLOP/s",
                            "frac": fl / (ms * 1e-3) / 1e12 / tail_peak, "avg_launch_us": ms * 1e3, "avg_launch_us_eager_sample": ms_raw * 1e3,
                            "duration_source": dur_note,
                            "algorithmic_flops_per_launch": fl, "traffic": None, "event_samples": len(ch.pairs),
                            "note": f"{per_row} flop per row x {rows:.0f} rows per step; avg_launch_us = the SUM of the tail's launches per step (each "
                                    "launch bracketed by its own HIP event pair); the per-launch split is in the rocprofv3 summary under profiles/"})
        dominant = max((k for k in kernels if not k["kernel"].startswith(("k_tail_gemm x", "library tail", "k_gemm_x x5", "k_gemm_h x5"))), key=lambda k: k["avg_launch_us"]) if kernels else None
        flops = {"cls": cfg.flops_cls(), "full": cfg.flops_full(), "clsfold": flops_clsfold(cfg)}[args.nn_path]
        # Boards the network really processed: eager stepping and the hand-written tail chain honour the live leaf count; only the
        # library tail (--tail library) runs the whole fixed-size leaf buffer every step.
        chain = (exact or (args.tail == "chain" and getattr(net, "chain_tail", False))) and args.nn_path == "clsfold"
        full_buffer_rows = sims_all * runner.leaves_per_step
        evals = leaves_all if (args.no_graph or chain) else full_buffer_rows
        nn_flop_alg = evals * flops                     # algorithmic flops of the function on the boards processed
        nn_flop_issued = nn_flop_alg
        if not args.no_graph and getattr(net, "fused_embed_pool", False) and args.nn_path == "clsfold":
            kreal_ = cfg.channels * cfg.patch_size ** 2
            front = 2 * (cfg.tokens - 1) * cfg.embed_dim * kreal_ + 2 * 2 * cfg.tokens * cfg.embed_dim * cfg.num_heads
            # the embedding / pooling kernel always honours the live count; the compacting kernel issues only `share` of its tiles
            share = next((k_["executed_share_of_algorithmic_flops"] for k_ in kernels if k_["kernel"].startswith(("k_embed_pool", "k_embed_fold"))), None) or 1.0
            nn_flop_alg = leaves_all * front + evals * (flops - front)
            nn_flop_issued = leaves_all * front * share + evals * (flops - front)
        out = {
            "metric": "selfplay_games_per_sec" if args.virtual_loss <= 1 else "selfplay_games_per_sec_virtual_loss", "value": games_per_s, "unit": "games/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt_max / args.steps * 1e3,
            "preroll": {"cheap_moves": args.preroll_cheap, "cheap_sims": args.preroll_sims, "full_moves": args.preroll_full,
                        "note": "untimed, independent of --warmup: de-phases the slots so the completion count is window-independent"},
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16" if args.nn_dtype == "bf16" else "f32", "data": "synthetic",
            "config": {"workload": f"Gomoku {args.size}x{args.size}, {args.sims} sims/move, {args.games} concurrent self-play games per GPU "
                                   f"(BASELINE.json configs[2]), continuous self-play", "games_per_gpu": args.games,
                       "sims_per_move": args.sims, "net": f"ViT patch5 embed{cfg.embed_dim} heads{cfg.num_heads} depth{cfg.depth} (ai/nn.py), random init seed 0, path={args.nn_path}, "
                                                           f"{args.nn_dtype}" + (f", tail={args.tail}" if args.nn_path == "clsfold" else "")
                                                           + (f", embed={args.embed}" if args.nn_dtype == "bf16" and args.nn_path == "clsfold" and cfg.depth == 1 else "")
                                                           + (f", full-token blocks={args.blocks}" if cfg.depth > 1 else ""),
                       "parallelism": f"games sharded over {world} GPU(s), no collectives on the generation path"},
            "sims_per_sec": sims_all / dt_max, "leaf_evals_per_sec": leaves_all / dt_max,
            "eval_cache": {"entries_per_game": args.cache_entries, "mode": args.cache, "hits_rank0": c.get("cache_hits", 0),
                           "hit_rate_rank0": c.get("cache_hits", 0) / max(1, c.get("cache_hits", 0) + c["leaves_evaluated"])},
            "nn_tflops_algorithmic": nn_flop_alg / dt_max / 1e12, "nn_tflops_issued": nn_flop_issued / dt_max / 1e12,
            "nn_flops_per_board": flops, "nn_boards_evaluated": evals,
            "moves": ("asynchronous: every game moves as soon as its own search is complete (k_move_async), at most "
                      f"{args.per_launch} simulation(s) per game and tree launch" + (f" while the launch is younger than {args.young_us} us" if args.young_us else "") + f"; a step = {args.games} moves of the batch in total") if args.async_moves
                     else "lock-step: all games of the batch move together",
            "stepping": "eager+sync" if args.no_graph else f"hipGraph replay, {runner.steps_per_graph} simulation step(s) per graph, {runner.n_split} game group(s) (per simulation: k_tree, the network kernels taking the pending leaves straight from the engine, no host sync)",
            "value_definition": value_src, "games_per_sec_renewal_estimate": est_games_per_s,
            "mean_plies_per_game": mean_plies, "game_length_source": src, "games_finished_in_window": fin_window,
            "plies_in_window": plies_all, "counters_rank0": c, "roofline": dominant, "roofline_puct": roof, "kernel_rooflines": kernels,
        }
        # measured IN THIS RUN: the fp32-accurate evaluator's own timed window (None when a side-line flag is set or --fp32-steps 0)
        out["fp32_line"] = fp32
        out["consistency"] = {"headline_window_s": dt_max, "fp32_window_s": fp32["seconds"] if fp32 else 0.0,
                              "timed_windows_sum_s": dt_max + (fp32["seconds"] if fp32 else 0.0),
                              "process_wall_s_so_far": time.perf_counter() - T_PROCESS_START,
                              "fits_in_driver_run": dt_max + (fp32["seconds"] if fp32 else 0.0) <= time.perf_counter() - T_PROCESS_START,
                              "note": "both timed windows lie inside this process; the driver's clock around the command must exceed their sum"}
        out["parity"] = {"tree_and_rules": "bit-exact vs the oracle / the reference's golden vectors (tests/test_gpu_engine.py); asserted by the -m gpu suite, not re-measured here"}
        try:        # what the evaluators do to the search results on the reference's recorded positions: QUOTED from a builder-run file
            par_file = next(f for f in ("r04_nn_parity.json", "r03_nn_parity.json") if os.path.exists(os.path.join(ROOT, "profiles", f)))
            par = json.load(open(os.path.join(ROOT, "profiles", par_file)))
            q = {"source": f"profiles/{par_file}",
                 "note": "builder-run measurement (tools/measure_nn_parity.py on a GPU box), quoted: these numbers do not move with this run",
                 "fp32_accurate_evaluator": {"logits_vs_reference_seed0": par["kat_vs_reference_seed0"].get("fp32_clsfold"),
                                             "visit_policy_vs_fp32_full": par["search_vs_fp32_full"].get("fp32_clsfold")}}
            if args.nn_dtype == "bf16":
                q["bf16_evaluator_vs_reference_fp32"] = par["kat_vs_reference_seed0"].get(f"bf16_{args.nn_path}")
                q["bf16_visit_policy_vs_fp32_evaluator"] = par["search_vs_fp32_full"].get("bf16_clsfold")
            elif not exact:
                q["torch_fp32_evaluator"] = par["search_vs_fp32_full"].get("fp32_cls")
            out["parity"]["quoted"] = q
        except Exception:
            pass
        out["tree_launches_per_move"] = launches / runner.n_split / args.steps
        if args.virtual_loss > 1:
            out["config"]["workload"] += f" - OPT-IN virtual-loss mode, {args.virtual_loss} leaves in flight per game (not the reference's sequential search)"
            out["virtual_loss"] = {"leaves_per_step": args.virtual_loss, "note": "separate metric key: results differ from the reference's search by design; "
                                   "the headline (no flag) is the parity mode"}
        if args.train_step:
            out["config"]["workload"] += " + one train step (batch 512 per GPU, fp32 autograd, fused gradient bucket all-reduce) after every move (BASELINE.json configs[4])"
            out["train_step"] = {"steps": len(train_ms), "ms_per_train_step": (sum(x.elapsed_time(y) for x, y in train_ms) / len(train_ms)) if train_ms else None,
                                 "batch_per_gpu": 512, "replay_tuples_rank0": replay.size(),
                                 "promotions_in_window": len(promote_ms), "promote_every_moves": args.promote_every,
                                 "ms_per_promotion": (sum(m for m, _ in promote_ms) / len(promote_ms)) if promote_ms else None,
                                 "promotions_in_place": sum(1 for _, ip in promote_ms if ip),
                                 "note": "promotion (main.py:55-59) = trained weights -> the self-play evaluator's device buffers, refreshed IN PLACE "
                                         "(folded constants and packed fragments recomputed on the device, same addresses: the captured step graphs "
                                         "are not re-captured) + eval-cache clear; its wall time is inside the timed window"}
        if args.cpu_seconds > 0 and world == 1:
            out["cpu_baseline"] = cpu_baseline(args.sims, args.cpu_seconds, mean_plies)
            out["gpu_over_cpu"] = games_per_s / out["cpu_baseline"]["value"]
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
