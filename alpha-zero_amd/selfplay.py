"""Batched self-play on the azk engine: thousands of <Game>.self_play loops as one lock-step batch.

Mirrors the per-move loop of games/gomoku.py:123-164 (and tictactoe.py:99-133, connect4.py:117-151):
fresh root -> n_sims simulations -> pi, q, raw board recorded -> sample (early moves) or most-visited
child -> make_move -> check_winner / draw.  Every game does that in the same step of the same kernels.

self_play_batch returns, per game, the tuple the reference's Gomoku.self_play returns
(boards, actions, policy_distributions, qs, winner) so train.collect_data's consumer
(train.save_data_to_buffer, train.py:30-49) can take it unchanged.
"""
import numpy as np

from azk import Engine

# gomoku.py:144 samples while move_count < 8; tictactoe.py:117 / connect4.py:135 sample every move
SAMPLE_UNTIL = {"gomoku": 8, "tictactoe": 1 << 30, "connect4": 1 << 30}


def cells_to_board(cells, planes, rows, cols, side_to_move):
    """int8 cell codes -> the reference's float32 board [F,R,C] (plane 2 = side to move, tictactoe.py:17,41)."""
    b = np.zeros((planes, rows, cols), np.float32)
    c = np.asarray(cells).reshape(rows, cols)
    b[0] = c == 1
    b[1] = c == 2
    if planes == 3:
        b[2] = side_to_move
    return b


class SelfPlayResult:
    __slots__ = ("boards", "actions", "pis", "qs", "winner", "cells", "replay_base")

    def __init__(self):
        self.boards, self.actions, self.pis, self.qs, self.winner, self.cells = [], [(-1, -1)], [], [], None, []
        self.replay_base = None      # first tuple index in the DeviceReplay stream (when a replay ring is attached)

    def as_reference_tuple(self):
        """(boards, actions, policy_distributions, qs, winner) - gomoku.py:164"""
        return self.boards, self.actions, self.pis, self.qs, self.winner


def self_play_batch(game, evaluator, n_games, n_sims, size=None, seed=0, first_global_game=0, dirichlet=True,
                    alpha=0.03, noise_fn=None, uniform_fn=None, device=0, leaf_dtype="float32", engine=None,
                    max_moves=None, sample_until=None, stats=None, replay=None, cache_entries=0, vanilla_rng=None, cache_shared=False,
                    budget_stepping=False, leaves_per_step=1):
    """Play n_games games to the end in one batch.

    evaluator(boards[n,F,R,C] CUDA) -> (logits [n,A], values [n] | [n,1]).
    RNG: by default Dirichlet noise and sampling uniforms come from the engine's counter-based generator
    keyed by (seed, first_global_game + g, move) - independent of how games are sharded over GPUs.
    noise_fn(move_idx) -> float64 [G, A] and uniform_fn(move_idx) -> float64 [G] (numpy) override it
    (that is how parity tests inject the reference's recorded np.random draws).
    An evaluator of None plays vanilla MCTS (random rollouts on the device, mcts.py:57-79) - for a whole game
    (self_play(None, n): greedy moves, tictactoe.py:117 / gomoku.py:146) or for one side of an (ev0, ev1) pair
    (test.compare(Game, None, model, ...), main.py:76).  vanilla_rng: uint32 [G, 625] MT19937 states (default: one
    np.random.RandomState per global game index derived from `seed`).
    """
    import torch
    max_sims = max(n_sims) if isinstance(n_sims, (tuple, list)) else n_sims
    eng = engine or Engine(game, n_games, max_sims, size=size, device=device, leaf_dtype=leaf_dtype, cache_entries=cache_entries,
                           cache_shared=cache_shared, leaves_per_step=leaves_per_step)
    budget_stepping = budget_stepping or eng.K > 1          # virtual-loss engines are driven by the simulation budget
    assert eng.G == n_games
    G, A = eng.G, eng.action_dim
    eng.reset_games()
    results = [SelfPlayResult() for _ in range(G)]
    active = np.ones(G, bool)
    su = SAMPLE_UNTIL[game] if sample_until is None else sample_until
    evs = list(evaluator) if isinstance(evaluator, (tuple, list)) else [evaluator]
    if any(e is None for e in evs):
        if vanilla_rng is None:
            from azk import mt_state_from_numpy
            vanilla_rng = np.stack([mt_state_from_numpy(np.random.RandomState(
                (int(seed) * 2654435761 + first_global_game + g) % (1 << 32)).get_state()) for g in range(G)])
        eng.vanilla_set_rng(vanilla_rng)
        if evaluator is None and sample_until is None:
            su = 0                                                # model=None: max_visit_child every move
    vanilla_chunk = 64 if eng.rows * eng.cols > 64 else None     # bounds one launch's run time on the big boards
    move = 0
    while active.any():
        if noise_fn is not None:
            nz = noise_fn(move)
            noise = torch.from_numpy(np.ascontiguousarray(nz, np.float64)).to(eng.device) if nz is not None else None
            uni = None
        else:
            noise, uni = eng.gen_noise(seed, first_global_game, move, alpha, want_noise=dirichlet)
        if uniform_fn is not None:
            uni = torch.from_numpy(np.ascontiguousarray(uniform_fn(move), np.float64)).to(eng.device)
        # an (evaluator0, evaluator1) pair plays the two sides (test.compete, test.py:79-84); all games are at the same ply
        ev = evaluator[move & 1] if isinstance(evaluator, (tuple, list)) else evaluator
        ns = n_sims[move & 1] if isinstance(n_sims, (tuple, list)) else n_sims
        if ev is None:
            eng.vanilla_search(ns, chunk=vanilla_chunk)
        elif budget_stepping:
            eng.search_budget(ev, ns, noise if dirichlet else None)
        else:
            eng.search(ev, ns, noise if dirichlet else None)
        pi, q, _ = eng.root_stats()
        cells_before, to_move, _ = eng.get_positions()
        chosen, winner, done = eng.advance(uni, su)
        bases = eng.emit_finished(replay).cpu().numpy() if replay is not None else None     # train.save_data_to_buffer on device
        pi_h, q_h = pi.cpu().numpy(), q.cpu().numpy()
        chosen_h, winner_h, done_h = chosen.cpu().numpy(), winner.cpu().numpy(), done.cpu().numpy()
        for g in np.nonzero(active)[0]:
            r = results[g]
            r.boards.append(cells_to_board(cells_before[g], eng.planes, eng.rows, eng.cols, to_move[g]))
            r.pis.append(pi_h[g].copy())
            r.qs.append(float(q_h[g]))
            c = int(chosen_h[g])
            r.cells.append(c)
            r.actions.append((c // eng.cols, c % eng.cols))
            if done_h[g]:
                r.winner = int(winner_h[g])
                active[g] = False
                if bases is not None:
                    r.replay_base = int(bases[g])
        move += 1
        if max_moves is not None and move >= max_moves:
            break
    eng.check_error()
    if stats is not None:
        stats.update(eng.counters())
        stats["moves"] = move
    return results


class _Half:
    """State of one independently stepped group of games (an engine + its static step buffers + pinned records)."""

    def __init__(self, torch, eng, dirichlet):
        self.eng = eng
        e = eng
        self.stats = torch.zeros(8, dtype=torch.int64, device=e.device)
        pin = dict(pin_memory=True)
        self.h_pi = torch.zeros((e.G, e.action_dim), dtype=torch.float64, **pin)
        self.h_q = torch.zeros(e.G, dtype=torch.float64, **pin)
        self.h_chosen = torch.zeros(e.G, dtype=torch.int32, **pin)
        self.h_winner = torch.zeros(e.G, dtype=torch.int32, **pin)
        self.h_done = torch.zeros(e.G, dtype=torch.int32, **pin)
        self.h_stats = torch.zeros(8, dtype=torch.int64, **pin)
        # static buffers so a captured step graph always sees the same addresses
        self.noise_buf = torch.zeros((e.G, e.action_dim), dtype=torch.float64, device=e.device) if dirichlet else None
        self.logits_buf = torch.zeros((e.slots, e.action_dim), dtype=torch.float32, device=e.device)
        self.values_buf = torch.zeros(e.slots, dtype=torch.float32, device=e.device)
        self.uni = None


def concurrent_streams(torch, device, n, spin_cycles=600_000):
    """n HIP streams whose kernels really run side by side.  HIP hands its streams out over a few hardware queues (four per priority by
    default) and two streams on one queue serialize - measured on this pool: of six fresh streams three pairs shared a queue, and the
    two streams the game groups used to get were such a pair (rocprofv3 kernel trace: same Queue_Id, 0.1 % of the time with two kernels
    in flight).  So candidates are probed pairwise with a one-thread spin kernel: two spins that take the time of one overlap."""
    import time
    cand = [torch.cuda.Stream(device=device) for _ in range(8)] + [torch.cuda.Stream(device=device, priority=-1) for _ in range(2)]
    for st in cand:                                  # queues are created at first use
        with torch.cuda.stream(st):
            torch.cuda._sleep(1000)
    torch.cuda.synchronize(device)

    def spins(a, b):
        torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        for st in (a, b):
            with torch.cuda.stream(st):
                torch.cuda._sleep(spin_cycles)
        torch.cuda.synchronize(device)
        return time.perf_counter() - t0
    serial = min(spins(cand[0], cand[0]) for _ in range(2))
    chosen = [cand[0]]
    for st in cand[1:]:
        if len(chosen) == n:
            break
        if all(min(spins(st, c) for _ in range(2)) < 0.75 * serial for c in chosen):
            chosen.append(st)
    if len(chosen) < n:
        if n == 2:
            # a noisy probe (another tenant on the box): a normal- and a high-priority stream never share a queue
            return [cand[0], cand[-1]]
        raise RuntimeError(f"only {len(chosen)} of {n} streams found that run concurrently on this device")
    return chosen


class SelfPlayRunner:
    """Continuous self-play: G slots advance one move per `play_move()`; a slot whose game ends restarts
    from an empty board in the same call (azk_recycle_finished), so every move does G searches.
    Per move the records the reference's self_play keeps (pi, q, action, winner; gomoku.py:138-146)
    are copied to pinned host buffers; `on_records` (optional) receives them.

    `n_split` groups of G/n_split games are stepped on separate HIP streams inside ONE captured graph: the
    latency-bound kernels of one group (tree walk, the cls-row tail) run under the bandwidth/VALU-bound kernels
    of the other.  Groups are independent games, so this changes no result.

    RNG key = (seed, first_global_game + slot, move counter): results do not depend on the sharding.
    """

    def __init__(self, game, evaluator, n_games, n_sims, size=None, seed=0, first_global_game=0, dirichlet=True,
                 alpha=0.03, device=0, leaf_dtype="float32", recycle=True, on_records=None, kernel_timer=None,
                 use_graph=False, n_split=1, replay=None, cache_entries=0, cache_shared=False, budget_stepping=False, per_launch=8,
                 steps_per_graph=32,
                 leaves_per_step=1):
        import torch
        self.replay = replay
        self.torch = torch
        # leaves_per_step > 1: OPT-IN virtual-loss expansion (K leaves in flight per game; changes search results); it is driven
        # by the simulation budget, like budget stepping
        self.leaves_per_step = max(1, int(leaves_per_step))
        self.budget_stepping, self.per_launch = (budget_stepping or self.leaves_per_step > 1) and use_graph, per_launch
        assert self.leaves_per_step == 1 or use_graph, "virtual-loss mode runs on the graph runner"
        self.launches = 0               # simulation-step launches issued (per game group) since construction
        self.use_graph = use_graph
        self._graph = None
        self.steps_per_graph = max(1, int(steps_per_graph))
        assert n_games % n_split == 0
        self.n_split = n_split if use_graph else 1
        per = n_games // self.n_split
        self.halves = [_Half(torch, Engine(game, per, n_sims, size=size, device=device, leaf_dtype=leaf_dtype,
                                           cache_entries=cache_entries, cache_shared=cache_shared, leaves_per_step=self.leaves_per_step), dirichlet)
                       for _ in range(self.n_split)]
        self.eng = self.halves[0].eng
        self.G = n_games
        self.evaluator, self.n_sims, self.seed, self.first = evaluator, n_sims, seed, first_global_game
        self.dirichlet, self.alpha, self.recycle, self.on_records = dirichlet, alpha, recycle, on_records
        self.sample_until = SAMPLE_UNTIL[game]
        self.kernel_timer = kernel_timer
        self.leaf_source_ok = True              # let a fused evaluator read the pending leaves straight from the engine (no azk_step_gather)
        self.streams = concurrent_streams(torch, self.eng.device, self.n_split) if self.n_split > 1 else []
        self.move_idx = 0
        self.plies_played = 0
        for h in self.halves:
            h.eng.reset_games()

    # ---- one move for every slot -------------------------------------------------------------------------
    def play_move(self):
        torch = self.torch
        per = self.halves[0].eng.G
        for i, h in enumerate(self.halves):
            noise, h.uni = h.eng.gen_noise(self.seed, self.first + i * per, self.move_idx, self.alpha, want_noise=self.dirichlet)
            if self.dirichlet:
                h.noise_buf.copy_(noise)
        if self.use_graph:
            self.search_graph()
        else:
            self.search(self.halves[0].noise_buf)
        for h in self.halves:
            e = h.eng
            pi, q, _ = e.root_stats()
            h.h_pi.copy_(pi, non_blocking=True)
            h.h_q.copy_(q, non_blocking=True)
            chosen, winner, done = e.advance(h.uni, self.sample_until)
            h.h_chosen.copy_(chosen, non_blocking=True)
            h.h_winner.copy_(winner, non_blocking=True)
            h.h_done.copy_(done, non_blocking=True)
            if self.replay is not None:
                e.emit_finished(self.replay)              # (state, pi, z) tuples with D4 augmentation, on device
            if self.recycle:
                e.recycle_finished(h.stats)
            h.h_stats.copy_(h.stats, non_blocking=True)
        torch.cuda.current_stream().synchronize()
        for i, h in enumerate(self.halves):
            self.plies_played += int((h.h_chosen >= 0).sum())
            if self.on_records is not None:
                self.on_records(self.move_idx, i * per, h.h_pi, h.h_q, h.h_chosen, h.h_winner, h.h_done)
        self.move_idx += 1

    def search(self, noise):
        """Eager stepping (host sync per simulation, n_leaf-sized evaluator batches); optional k_tree event timing."""
        e, torch, kt = self.eng, self.torch, self.kernel_timer
        e.begin_search(noise)
        logits = values = None
        for s in range(self.n_sims):
            if kt is not None and kt.want(s):
                kt.start()
                e.step_tree(logits, values)
                kt.stop()
                e.step_gather()
            else:
                e.step(logits, values)
            n = int(e.n_leaf.item())
            if n > 0:
                logits, values = self.evaluator(e.leaf_boards[:n])
                logits = logits.to(torch.float32).contiguous()
                values = values.to(torch.float32).reshape(-1).contiguous()
            elif e.cache_entries:
                logits, values = e._no_logits, e._no_values
            else:
                logits = values = None
        if logits is not None:
            e.step_expand_backup(logits, values)

    def _step_body(self, h, timer=None):
        """One simulation for every game of one group, with no host round trip: expand+backup of the previous leaves,
        PUCT select, leaf compaction, then the evaluator over the (fixed-size) leaf buffer.  Rows past n_leaf hold
        older boards; kernels that honour `live_count` skip them and nothing ever reads their outputs."""
        e = h.eng
        from_leaves = getattr(self.evaluator, "fused_embed_pool", False) and self.leaf_source_ok
        if hasattr(self.evaluator, "leaf_source"):
            self.evaluator.leaf_source = e.leaf_source() if from_leaves else None
        if timer is not None:
            timer.start()
            e.step_tree(h.logits_buf, h.values_buf)
            timer.stop()
            if not from_leaves:
                e.step_gather()
        elif from_leaves:
            e.step_tree(h.logits_buf, h.values_buf)          # the network kernel compacts the leaves itself
        else:
            e.step(h.logits_buf, h.values_buf)
        had_fast = getattr(self.evaluator, "fast_outputs", False)
        if hasattr(self.evaluator, "live_count"):
            self.evaluator.live_count = e.n_leaf
            self.evaluator.fast_outputs = True
        if hasattr(self.evaluator, "kernel_timers"):
            if timer is None:
                self.evaluator.kernel_timers = None
            elif getattr(self.evaluator, "fused_embed_pool", False):
                self.evaluator.kernel_timers = (timer.child("k_embed_pool"), timer.child("k_tail"))
            else:
                self.evaluator.kernel_timers = (timer.child("k_embed"), timer.child("k_cls_pool"), timer.child("k_tail"))
        if hasattr(self.evaluator, "out_buffers"):
            self.evaluator.out_buffers = (h.logits_buf, h.values_buf)
        logits, values = self.evaluator(e.leaf_boards)
        if logits.data_ptr() != h.logits_buf.data_ptr():        # evaluators that do not write the step buffers themselves
            h.logits_buf.copy_(logits)
            h.values_buf.copy_(values.reshape(-1))
        if hasattr(self.evaluator, "out_buffers"):
            self.evaluator.out_buffers = None
        if hasattr(self.evaluator, "leaf_source"):
            self.evaluator.leaf_source = None
        if hasattr(self.evaluator, "live_count"):
            # the step's row count belongs to THIS engine: an evaluator shared with another runner (or called directly afterwards)
            # must not keep honouring it
            self.evaluator.live_count = None
            self.evaluator.fast_outputs = had_fast

    def _all_bodies(self):
        torch = self.torch
        if self.n_split == 1:
            self._step_body(self.halves[0])
            return
        cur = torch.cuda.current_stream()
        for st in self.streams:
            st.wait_stream(cur)
        for h, st in zip(self.halves, self.streams):
            with torch.cuda.stream(st):
                self._step_body(h)
        for st in self.streams:
            cur.wait_stream(st)

    def search_graph(self):
        """Replays of captured hipGraphs (per group: k_tree -> evaluator kernels).  Each group has its own graph, replayed on its
        own stream.  One-simulation stepping replays n_sims times; budget stepping (games run on inside a launch while their
        simulations need no evaluator) replays until no game owes simulations - about 45 % of n_sims at the benchmark config."""
        torch = self.torch
        for h in self.halves:
            if self.budget_stepping:
                h.eng.begin_search_budget(h.noise_buf, self.n_sims, self.per_launch)
            else:
                h.eng.begin_search(h.noise_buf)
        cur = torch.cuda.current_stream()
        if self._graph is None:
            # the first simulations run eagerly on a side stream (allocator warm-up), the capture records one more
            side = torch.cuda.Stream()
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                for _ in range(3):
                    for h in self.halves:
                        self._step_body(h)
            cur.wait_stream(side)
            torch.cuda.synchronize()
            self._graph, self._graph_many = [], []
            for h in self.halves:
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    self._step_body(h)
                self._graph.append(g)
                if self.steps_per_graph > 1:
                    # the same step `steps_per_graph` times in one graph: consecutive graph launches leave the GPU idle for ~8 us
                    # (rocprofv3 kernel trace: gap before k_tree), consecutive kernels of one graph do not
                    gm = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(gm):
                        for _ in range(self.steps_per_graph):
                            self._step_body(h)
                    self._graph_many.append(gm)
            done = 3                     # the three eager steps ran; the captured ones were only recorded
        else:
            done = 0
        kt = self.kernel_timer
        streams = self.streams if self.n_split > 1 else [cur]
        for st in self.streams:
            st.wait_stream(cur)

        def replay(s, timed=None, many=False):
            timed = (kt is not None and kt.want(s)) if timed is None else timed
            for i, (h, g, st) in enumerate(zip(self.halves, self._graph, streams)):
                with torch.cuda.stream(st):
                    if timed:
                        # sampled steps run the same kernels eagerly so HIP events can bracket k_tree on its stream
                        self._step_body(h, timer=kt)
                    elif many:
                        self._graph_many[i].replay()
                    else:
                        g.replay()
        if not self.budget_stepping:
            # (the sampling phase carries over from move to move: with a stride that does not divide n_sims the sampled simulations
            #  drift through the search - leaf counts differ between its early and late simulations - instead of hitting the same few)
            s, U, next_sample = done, self.steps_per_graph, max(done, getattr(self, "_sample_phase", 0))
            while s < self.n_sims:
                if kt is not None and kt.enabled and s >= next_sample and len(kt.pairs) < kt.max:
                    replay(s, timed=True)
                    next_sample = s + kt.stride
                    s += 1
                elif U > 1 and s + U <= self.n_sims:
                    replay(s, timed=False, many=True)
                    s += U
                else:
                    replay(s, timed=False)
                    s += 1
            self.launches += self.n_sims
            if kt is not None:
                self._sample_phase = max(0, next_sample - self.n_sims)
        else:
            # no game can finish before its budget's worth of cache misses: a first stretch without looking, then a look (one
            # 4-byte read-back) every few launches
            s, first = done, max(done, int(0.36 * self.n_sims / self.leaves_per_step))
            while True:
                stop = first if s < first else s + 8
                while s < stop:
                    replay(s)
                    s += 1
                for st in self.streams:
                    cur.wait_stream(st)
                if all(h.eng.unfinished() == 0 for h in self.halves):
                    break
                if s > 2 * self.n_sims + 16:
                    raise RuntimeError("budget stepping does not terminate")
            self.launches += s
        if self.leaves_per_step == 1:            # (K > 1: the loop above only ends once nothing is pending)
            for h, st in zip(self.halves, streams):
                with torch.cuda.stream(st):
                    h.eng.step_expand_backup(h.logits_buf, h.values_buf)
        for st in self.streams:
            cur.wait_stream(st)

    @property
    def games_finished(self):
        return sum(int(h.h_stats[0]) for h in self.halves)

    @property
    def finished_plies(self):
        return sum(int(h.h_stats[1]) for h in self.halves)

    def counters(self):
        tot = {}
        for h in self.halves:
            for k, v in h.eng.counters().items():
                tot[k] = tot.get(k, 0) + v
        return tot

    def reset_counters(self):
        for h in self.halves:
            h.eng.reset_counters()

    def check_error(self):
        for h in self.halves:
            h.eng.check_error()


class AsyncSelfPlayRunner:
    """Continuous self-play with ASYNCHRONOUS moves (games/gomoku.py:132-162: a game moves as soon as ITS search is done - in the
    batched engine: no game waits for the slowest search of the batch) and budget stepping (a game keeps simulating inside a tree
    launch while its simulations need no evaluator - terminal leaves, eval-cache hits -, at most `per_launch` simulations per
    launch).  Trees, moves and games are those of the lock-step runner, slot for slot and move for move: a game's simulations stay
    sequential, the random keys are (seed, global game, the slot's move counter) (tests/test_gpu_async.py).

    One step = k_tree (budget-stepped) -> k_move_async (moves every game whose search is complete, starts its next search) ->
    evaluator over the pending leaves; `steps_per_graph` steps are captured in one hipGraph.  After every graph replay the host
    enqueues the drain (finished games: (state, pi, z) emission into `replay`, statistics, restart) and an asynchronous copy of
    the statistics; it looks at them one replay late, so the GPU never waits for the host.

    play_move() keeps the lock-step runner's contract for its callers (bench.py): it returns once the batch has played G more
    moves in total (one move per resident game on average) - G x n_sims simulations, the same work as one lock-step move."""

    def __init__(self, game, evaluator, n_games, n_sims, size=None, seed=0, first_global_game=0, dirichlet=True, alpha=0.03, device=0,
                 leaf_dtype="float32", recycle=True, on_records=None, kernel_timer=None, replay=None, cache_entries=0, cache_shared=False,
                 per_launch=2, steps_per_graph=32, record_capacity=None, use_graph=True, young_launch_us=0):
        import torch
        self.torch, self.replay, self.evaluator = torch, replay, evaluator
        self.eng = Engine(game, n_games, n_sims, size=size, device=device, leaf_dtype=leaf_dtype, cache_entries=cache_entries, cache_shared=cache_shared)
        e = self.eng
        self.G, self._n_sims, self.n_split, self.leaves_per_step = n_games, n_sims, 1, 1
        self.per_launch, self.steps_per_graph, self.use_graph = int(per_launch), max(1, int(steps_per_graph)), use_graph
        self.kernel_timer, self.on_records = kernel_timer, on_records
        self.halves = [_Half(torch, e, False)]              # (logits / values step buffers; .eng for callers that walk the groups)
        self.h = self.halves[0]
        self.leaf_source_ok = True
        e.reset_games()
        cap = (4 * n_games if record_capacity is None else record_capacity) if (on_records is not None or record_capacity) else 0
        self.stats, self.records = e.async_begin(n_sims, self.per_launch, SAMPLE_UNTIL[game], seed, first_global_game, alpha, dirichlet, recycle, cap,
                                                 young_launch_us=young_launch_us)
        self.rec_cap, self.rec_read, self.copy_stream = cap, 0, None
        self.h_stats = [torch.zeros(16, dtype=torch.int64, pin_memory=True) for _ in range(2)]
        self.events = [None, None]
        self.launches = 0               # simulation-step launches issued
        self.chunks = 0
        self.move_target = 0
        self._graph = None
        self._seen = torch.zeros(16, dtype=torch.int64)
        self.move_idx = 0

    @property
    def n_sims(self):
        return self._n_sims

    @n_sims.setter
    def n_sims(self, v):
        """Simulations per search from now on (bench.py's cheap pre-roll); read from device memory by the captured graphs."""
        if int(v) != self._n_sims:
            self._n_sims = int(v)
            self.eng.async_set_budget(self._n_sims, self.per_launch)

    # the lock-step runner's step body, with the asynchronous tree step in front
    def _step_body(self, timer=None):
        h, e, ev = self.h, self.eng, self.evaluator
        from_leaves = getattr(ev, "fused_embed_pool", False) and self.leaf_source_ok
        if hasattr(ev, "leaf_source"):
            ev.leaf_source = e.leaf_source() if from_leaves else None
        if timer is not None:
            timer.start()
            e.async_step(h.logits_buf, h.values_buf, 1)     # k_tree alone between the events
            timer.stop()
            e.async_step(h.logits_buf, h.values_buf, 2)
        else:
            e.async_step(h.logits_buf, h.values_buf, 3)
        if not from_leaves:
            e.step_gather()
        had_fast = getattr(ev, "fast_outputs", False)
        if hasattr(ev, "live_count"):
            ev.live_count, ev.fast_outputs = e.n_leaf, True
        if hasattr(ev, "kernel_timers"):
            ev.kernel_timers = None if timer is None else ((timer.child("k_embed_pool"), timer.child("k_tail")) if getattr(ev, "fused_embed_pool", False)
                                                            else (timer.child("k_embed"), timer.child("k_cls_pool"), timer.child("k_tail")))
        if hasattr(ev, "out_buffers"):
            ev.out_buffers = (h.logits_buf, h.values_buf)
        logits, values = ev(e.leaf_boards)
        if logits.data_ptr() != h.logits_buf.data_ptr():
            h.logits_buf.copy_(logits)
            h.values_buf.copy_(values.reshape(-1))
        if hasattr(ev, "out_buffers"):
            ev.out_buffers = None
        if hasattr(ev, "leaf_source"):
            ev.leaf_source = None
        if hasattr(ev, "live_count"):
            ev.live_count, ev.fast_outputs = None, had_fast

    def _capture(self):
        torch = self.torch
        cur = torch.cuda.current_stream()
        side = torch.cuda.Stream()
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            for _ in range(3):                              # allocator warm-up; these steps count
                self._step_body()
        cur.wait_stream(side)
        torch.cuda.synchronize()
        self.launches += 3
        self._graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self._graph):
            for _ in range(self.steps_per_graph):
                self._step_body()

    def run_chunk(self):
        """`steps_per_graph` steps, the drain, and an asynchronous copy of the statistics (read one chunk late)."""
        torch = self.torch
        if not self.use_graph:
            for _ in range(self.steps_per_graph):
                self._step_body()
        else:
            if self._graph is None:
                self._capture()
            kt = self.kernel_timer
            if kt is not None and kt.enabled and len(kt.pairs) < kt.max and self.chunks % max(1, kt.stride // self.steps_per_graph) == 0:
                self._step_body(timer=kt)                   # a sampled step runs eagerly so HIP events can bracket its kernels
                self.launches += 1
            self._graph.replay()
        self.launches += self.steps_per_graph
        self.eng.async_drain(self.replay)
        i = self.chunks & 1
        self.h_stats[i].copy_(self.stats, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self.events[i] = ev
        self.chunks += 1

    def _look(self, i):
        """Statistics of the chunk before last (its copy has certainly landed once its event is done)."""
        if self.events[i] is not None:
            self.events[i].synchronize()
            self._seen = self.h_stats[i].clone()
            self._deliver_records()

    def _deliver_records(self):
        if self.on_records is None or self.records is None:
            return
        cursor = int(self._seen[6])
        if cursor - self.rec_read > self.rec_cap:
            raise RuntimeError("move-record ring overrun: raise record_capacity or drain more often")
        if cursor == self.rec_read:
            return
        # on a stream of their own: the entries are complete (their chunk's event is done) and the copies must not queue behind the
        # chunk that is running now
        if self.copy_stream is None:
            self.copy_stream = self.torch.cuda.Stream(device=self.eng.device)
        with self.torch.cuda.stream(self.copy_stream):
            idx = self.torch.arange(self.rec_read, cursor, device=self.eng.device) % self.rec_cap
            meta, q, pi = self.records["meta"][idx].cpu().numpy(), self.records["q"][idx].cpu().numpy(), self.records["pi"][idx].cpu().numpy()
        self.rec_read = cursor
        self.on_records(meta, q, pi)

    def run_until_moves(self, total_moves):
        """Run until the batch has played `total_moves` moves since construction (all slots together)."""
        while int(self._seen[5]) < total_moves:
            self.run_chunk()
            self._look((self.chunks & 1))                   # the OTHER buffer: the chunk before the one just enqueued
        return int(self._seen[5])

    def play_move(self):
        self.move_target += self.G
        self.run_until_moves(self.move_target)
        self.move_idx += 1

    def finish(self):
        """Wait for everything enqueued and read the final statistics."""
        self.torch.cuda.synchronize()
        self._seen = self.stats.cpu()
        self._deliver_records()
        return self._seen

    @property
    def plies_played(self):
        return int(self._seen[5])

    @property
    def games_finished(self):
        return int(self._seen[0])

    @property
    def finished_plies(self):
        return int(self._seen[1])

    def counters(self):
        return self.eng.counters()

    def reset_counters(self):
        self.eng.reset_counters()

    def check_error(self):
        self.eng.check_error()


class KernelTimer:
    """HIP-event timing of one kernel on the stream it is launched on (torch's current stream)."""

    def __init__(self, stride=16, max_samples=4096):
        import torch
        self.torch, self.stride, self.max = torch, stride, max_samples
        self.pairs = []
        self.enabled = False
        self.children = {}                       # named sub-timers sharing this timer's sampling decisions

    def child(self, name):
        if name not in self.children:
            self.children[name] = KernelTimer(self.stride, self.max)
        return self.children[name]

    def want(self, step):
        return self.enabled and step % self.stride == 0 and len(self.pairs) < self.max

    def start(self):
        self._a = self.torch.cuda.Event(enable_timing=True)
        self._a.record()

    def stop(self):
        b = self.torch.cuda.Event(enable_timing=True)
        b.record()
        self.pairs.append((self._a, b))

    def mean_ms(self):
        self.torch.cuda.synchronize()
        if not self.pairs:
            return None
        return sum(a.elapsed_time(b) for a, b in self.pairs) / len(self.pairs)

    def robust_mean_ms(self):
        """Mean of the samples that are not host stalls: a pair brackets an EAGER launch, and a host hiccup between its two event records
        (garbage collection, another process tearing down) adds milliseconds to a ~10 us sample - a handful of those moves the mean of a
        few hundred samples by tens of percent.  Samples above three times the median are dropped; returns (mean_ms, dropped, total)."""
        self.torch.cuda.synchronize()
        if not self.pairs:
            return None, 0, 0
        t = sorted(a.elapsed_time(b) for a, b in self.pairs)
        med = t[len(t) // 2]
        kept = [x for x in t if x <= 3.0 * med]
        return sum(kept) / len(kept), len(t) - len(kept), len(t)

    def spread_us(self):
        """(median, max) of the samples in microseconds: a mean that sits far above the median is a few stalled samples, not the kernel."""
        self.torch.cuda.synchronize()
        t = sorted(a.elapsed_time(b) * 1e3 for a, b in self.pairs)
        return (t[len(t) // 2], t[-1]) if t else (None, None)
