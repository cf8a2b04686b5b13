#!/usr/bin/env python3
"""bench.py - self-play data generation throughput on MI355X (BASELINE.json metric).

  python bench.py [--gpus N] [--steps K] [--warmup W]            (N > 1: launched by torch.distributed.run)

Workload (BASELINE.json configs[2]): Gomoku 15x15, 800 simulations per move, 2048 concurrent self-play
games per GPU, policy-value net = the reference's training config (ai/nn.py Net(15, patch 5, embed 512,
225 actions, 8 heads, depth 1, 2 channels), random init, seed 0), synthetic start (empty boards).

A STEP is one move of the whole resident batch: 2048 searches x 800 simulations (tree kernels + leaf
compaction + network evaluation + expansion/backup), move selection, state advance, the per-move records
(board, pi, q, action) copied to the host, and finished games restarted (continuous self-play).
value = self-play games/sec = games COMPLETED inside the timed region / seconds.  All games start in phase, so the
default warm-up (40 moves, about two game lengths) lets the phases decorrelate before timing; the renewal estimate
(plies per second / mean game length) is reported next to it and is used only when the window is too short to count.

One JSON line on rank 0.  Extra objects:
  roofline      k_tree (PUCT scan + expand + backup), HBM-bound: algorithmic bytes per launch from the engine's
                device counters / mean launch duration from HIP events on the launch stream
  cpu_baseline  the oracle (CPU restatement of the reference algorithm, batch-1 fp32 net, eval cache) timed on
                this box's host cores on a bounded sample of the same workload
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "alpha-zero_amd"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

MFMA_BF16_PEAK_TFLOPS = 2500.0      # dense bf16 MFMA peak, MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
BF16_PEAK_TFLOPS = 2500.0      # dense bf16 MFMA


def algorithmic_bytes(c, action_dim):
    """k_tree traffic model (DESIGN.md 'k_tree roofline'): per scanned child N,W,P = 4+8+4 B; per backed-up
    node a read-modify-write of N and W = 2*(4+8) B; per created child N,W,P,cell,first_child,n_children =
    4+8+4+2+4+2 B plus its move-list entry written then read (2+2 B); per evaluated leaf the logits row (4*A B),
    the leaf board written (R*C B) and the path written then read (8 B per node, folded into the trace term)."""
    return (16 * c["edges_scanned"] + 24 * c["trace_nodes"] + 28 * c["edges_created"]
            + c["leaves_evaluated"] * (4 * action_dim + action_dim) + 8 * c["trace_nodes"])


def cpu_baseline(n_sims, budget_s, mean_plies):
    """The oracle (kind 'port'): sequential search in C, batch-1 float32 ViT through PyTorch CPU, eval cache on,
    numpy softmax - the reference's algorithm and evaluator shape.  Bounded by wall-clock."""
    import numpy as np
    import torch
    from oracle import az_oracle as ao
    from pvnet import NetConfig, PolicyValueNet
    # batch-1 evaluation does not scale past a few threads (128 threads measured 45-64 sims/s on this class of
    # host, 8 threads several times that); 8 is also what the survey's reference measurement used
    torch.set_num_threads(min(8, os.cpu_count() or 1))
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
    out = ao.self_play(game, ev, n_sims, noise_fn=lambda mc: rng.dirichlet([0.03] * 225),
                       uniform_fn=lambda mc: rng.random_sample(), cache=cache, counters=cnt, time_budget=budget_s)
    dt = time.time() - t0
    sims_per_s = cnt.mcts_count / dt
    games_per_s = sims_per_s / (n_sims * mean_plies)
    return {"value": games_per_s, "unit": "games/s", "cores": torch.get_num_threads(), "kind": "port",
            "sims_per_sec": sims_per_s, "host_cpus": os.cpu_count(),
            "sample": f"first {len(out['cells'])} moves of one Gomoku 15x15 game, {n_sims} sims/move: {cnt.mcts_count} sims "
                      f"({cnt.evals} net evals, {cnt.matched} cache hits) in {dt:.1f} s; games/s = sims/s / ({n_sims} x {mean_plies:.1f} plies)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=40)
    ap.add_argument("--games", type=int, default=2048, help="concurrent games per GPU")
    ap.add_argument("--sims", type=int, default=800)
    ap.add_argument("--size", type=int, default=15)
    ap.add_argument("--nn-path", default="clsfold", choices=["clsfold", "cls", "full"])
    ap.add_argument("--no-graph", action="store_true", help="eager stepping with a host sync per simulation (n_leaf-sized batches)")
    ap.add_argument("--split", type=int, default=1, help="independent game groups stepped on separate streams inside the step graph")
    ap.add_argument("--cache-entries", type=int, default=32768, help="per-game eval-cache entries (MCTS.cache; 64 GB of HBM at 2048 games x 32768); 0 = off")
    ap.add_argument("--cpu-seconds", type=float, default=20.0, help="wall-clock budget of the CPU baseline leg (0 = skip)")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--train-step", action="store_true",
                    help="BASELINE.json configs[4]: one train.py step (batch 512 per GPU, Adam lr 2.5e-4, one fused gradient bucket "
                         "all-reduced over RCCL) after every move, fed from the device-resident replay ring")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from shard import env_world, shard_range
    rank, local_rank, world = env_world()
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    assert world == args.gpus or world == 1, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    torch.cuda.set_device(local_rank)
    torch.set_num_threads(min(8, os.cpu_count() or 1))       # the host only launches kernels; N ranks must not each spawn 128 threads

    from pvnet import NetConfig, PolicyValueNet
    from selfplay import KernelTimer, SelfPlayRunner
    A = args.size * args.size
    cfg = NetConfig(args.size, args.size, 2, A, 5, 512, 8, 1)
    net = PolicyValueNet(cfg, seed=0, device=f"cuda:{local_rank}", dtype=torch.bfloat16, path=args.nn_path)
    kt = KernelTimer(stride=16)
    replay = trainer = None
    if args.train_step:
        from azk import DeviceReplay
        from trainer import Trainer
        replay = DeviceReplay(400000, cfg.channels, cfg.rows, cfg.cols, cfg.action_dim, device=torch.device("cuda", local_rank))
        trainer = Trainer(cfg, net.state_dict(), device=f"cuda:{local_rank}")
    runner = SelfPlayRunner("gomoku", net, args.games, args.sims, size=args.size, seed=args.seed,
                            first_global_game=shard_range(args.games, rank)[0], device=local_rank, leaf_dtype="bfloat16",
                            recycle=True, kernel_timer=kt, use_graph=not args.no_graph, n_split=args.split, cache_entries=args.cache_entries,
                            replay=replay)
    train_ms = []

    def train_one():
        """train.train with one iteration (train.py:85-123): fresh Adam, the reference's loss, gradient bucket all-reduce."""
        if replay.size() < 512:
            return
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        trainer.train([replay.sample(512)], 0.00025, dist=dist if world > 1 else None)
        b.record()
        train_ms.append((a, b))
    eng = runner.eng

    def sync_all():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        runner.play_move()
        if args.train_step:
            train_one()
    train_ms.clear()
    runner.reset_counters()
    plies0, fin0, finp0 = runner.plies_played, runner.games_finished, runner.finished_plies
    kt.enabled = True
    sync_all()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        runner.play_move()
        if args.train_step:
            train_one()
    sync_all()
    dt = time.perf_counter() - t0
    kt.enabled = False
    runner.check_error()
    c = runner.counters()
    plies = runner.plies_played - plies0
    fin, finp = runner.games_finished, runner.finished_plies

    # max time / summed work over ranks (the only collectives: measurement, never the generation path)
    from shard import reduce_measurement
    dt_max, (plies_all, fin_window, fin_all, finp_all, sims_all, leaves_all) = reduce_measurement(
        dt, [plies, fin - fin0, fin, finp, c["sims"], c["leaves_evaluated"]], dist if world > 1 else None, "cuda")

    if rank == 0:
        if fin_all > 0:
            mean_plies, src = finp_all / fin_all, f"{int(fin_all)} games finished in this run"
        else:
            mean_plies, src = 30.0, "no game finished in this run: assumed 30 plies"
        est_games_per_s = plies_all / mean_plies / dt_max          # renewal estimate: plies per second / mean game length
        if fin_window >= 0.25 * args.games * world:
            games_per_s, value_src = fin_window / dt_max, "games completed inside the timed window / seconds"
        else:   # window too short for completions to be meaningful (all games start in phase): fall back to the estimate
            games_per_s, value_src = est_games_per_s, "plies in window / mean plies per finished game / seconds (window too short for a direct count)" 
        tree_ms = kt.mean_ms()
        launches = args.steps * args.sims * runner.n_split      # k_tree launches (one per game group per simulation)
        alg_bytes = algorithmic_bytes(c, A) / launches
        roof = None
        traffic, traffic_src, pmc = None, None, {}
        try:        # HBM bytes per k_tree launch from the PMC passes committed under profiles/ (FETCH_SIZE + WRITE_SIZE)
            pmc = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")))
            traffic = pmc["k_tree<true, true>"]["bytes_per_launch"]
            traffic_src = "profiles/r01_pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, same workload)"
        except Exception:
            pass
        if tree_ms:
            gbs = alg_bytes / (tree_ms * 1e-3) / 1e9
            roof = {"kernel": "k_tree<expand,select>", "bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": gbs / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src, "avg_launch_us": tree_ms * 1e3,
                    "algorithmic_bytes_per_launch": alg_bytes, "event_samples": len(kt.pairs)}
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
            ms = ch.mean_ms() if ch else None
            if ms:
                by = per_board * live
                kernels.append({"kernel": name, "bound": "hbm", "achieved": by / (ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                "frac": by / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "avg_launch_us": ms * 1e3, "algorithmic_bytes_per_launch": by,
                                "traffic": None, "event_samples": len(ch.pairs),
                                "note": f"{per_board} B per live board x {live:.0f} live boards per launch (device-side count)"})
        ch = kt.children.get("k_embed_pool")            # fused embedding + cls pooling: on-chip, priced against the dense bf16 MFMA peak
        ms = ch.mean_ms() if ch else None
        if ms:
            kreal = cfg.channels * cfg.patch_size ** 2
            per_board = 2 * (T_tok - 1) * Dm * kreal + 2 * T_tok * 16 * kreal + 2 * T_tok * cfg.num_heads * Dm
            fl = per_board * live
            ep_traffic = None
            try:
                ep_traffic = next(v["bytes_per_launch"] for k_, v in pmc.items() if k_.startswith("k_embed_pool"))
            except Exception:
                pass
            kernels.append({"kernel": "k_embed_pool", "bound": "mfma", "achieved": fl / (ms * 1e-3) / 1e12, "peak": MFMA_BF16_PEAK_TFLOPS,
                            "unit": "TFLOP/s", "frac": fl / (ms * 1e-3) / 1e12 / MFMA_BF16_PEAK_TFLOPS, "avg_launch_us": ms * 1e3,
                            "algorithmic_flops_per_launch": fl, "traffic": ep_traffic,
                            "traffic_source": traffic_src if ep_traffic else None, "event_samples": len(ch.pairs),
                            "note": f"{per_board} flop per live board (conv + score columns + weighted token sum) x {live:.0f} live boards per launch; "
                                    "HBM traffic is 9 KB per board (board in, z out): the kernel is bound by VALU/MFMA issue and LDS, not HBM"})
        dominant = max(kernels, key=lambda k: k["avg_launch_us"]) if kernels else None
        flops = {"cls": cfg.flops_cls(), "full": cfg.flops_full(), "clsfold": flops_clsfold(cfg)}[args.nn_path]
        evals = leaves_all if args.no_graph else sims_all      # graph mode runs the cls-row tail over the full fixed-size leaf buffer every step
        nn_flop_total = evals * flops
        if not args.no_graph and getattr(net, "fused_embed_pool", False) and args.nn_path == "clsfold":
            # ... but the embedding / pooling kernel honours the live leaf count: only the tail is paid for dead rows
            kreal_ = cfg.channels * cfg.patch_size ** 2
            front = 2 * (cfg.tokens - 1) * cfg.embed_dim * kreal_ + 2 * 2 * cfg.tokens * cfg.embed_dim * cfg.num_heads
            nn_flop_total = leaves_all * front + sims_all * (flops - front)
        out = {
            "metric": "selfplay_games_per_sec", "value": games_per_s, "unit": "games/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt_max / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": f"Gomoku {args.size}x{args.size}, {args.sims} sims/move, {args.games} concurrent self-play games per GPU "
                                   f"(BASELINE.json configs[2]), continuous self-play", "games_per_gpu": args.games,
                       "sims_per_move": args.sims, "net": f"ViT patch5 embed512 heads8 depth1 (ai/nn.py), random init seed 0, path={args.nn_path}",
                       "parallelism": f"games sharded over {world} GPU(s), no collectives on the generation path"},
            "sims_per_sec": sims_all / dt_max, "leaf_evals_per_sec": leaves_all / dt_max,
            "eval_cache": {"entries_per_game": args.cache_entries, "hits_rank0": c.get("cache_hits", 0),
                           "hit_rate_rank0": c.get("cache_hits", 0) / max(1, c.get("cache_hits", 0) + c["leaves_evaluated"])},
            "nn_tflops_executed": nn_flop_total / dt_max / 1e12, "nn_flops_per_board": flops, "nn_boards_evaluated": evals,
            "stepping": "eager+sync" if args.no_graph else f"hipGraph replay, {runner.n_split} game group(s) (per simulation: k_tree, the network kernels taking the pending leaves straight from the engine, no host sync)",
            "value_definition": value_src, "games_per_sec_renewal_estimate": est_games_per_s,
            "mean_plies_per_game": mean_plies, "game_length_source": src, "games_finished_in_window": fin_window,
            "plies_in_window": plies_all, "counters_rank0": c, "roofline": dominant, "roofline_puct": roof, "kernel_rooflines": kernels,
        }
        if args.train_step:
            out["config"]["workload"] += " + one train step (batch 512 per GPU, fp32 autograd, fused gradient bucket all-reduce) after every move (BASELINE.json configs[4])"
            out["train_step"] = {"steps": len(train_ms), "ms_per_train_step": (sum(x.elapsed_time(y) for x, y in train_ms) / len(train_ms)) if train_ms else None,
                                 "batch_per_gpu": 512, "replay_tuples_rank0": replay.size(),
                                 "note": "the network weights used for self-play are not refreshed inside the timed window (promotion happens per iteration in train_loop.py)"}
        if args.cpu_seconds > 0 and world == 1:
            out["cpu_baseline"] = cpu_baseline(args.sims, args.cpu_seconds, mean_plies)
            out["gpu_over_cpu"] = games_per_s / out["cpu_baseline"]["value"]
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
