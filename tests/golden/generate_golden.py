#!/usr/bin/env python3
"""Generate golden vectors by RUNNING THE REFERENCE (read-only at /root/reference).

TEST INFRASTRUCTURE.  Run in the build container only (the reference never travels
to the GPU box):   python3 -B tests/golden/generate_golden.py
Outputs: tests/golden/*.npz (data only: inputs + the reference's outputs).

The reference has no tests / golden vectors of its own, so every
fixture here comes from importing its modules and calling its functions:
  games/{tictactoe,connect4,gomoku}.py  statics (rules)            -> rules_*.npz
  ai/mcts.py MCTS.mcts + ai/node.py + utils.py (search)            -> search.npz
  <Game>.self_play + train.save_data_to_buffer (whole games)       -> games.npz
  ai/nn.py Net (small config, committed weights)                   -> nn_small.npz

Import hygiene: `import games` would create <reference>/logs/*.log through
utils.get_game_logger at class-definition time (utils.py:71-90).  We neutralise the
FileHandler and makedirs BEFORE importing so nothing is written under /root/reference,
and disable bytecode writing.  Gomoku 15x15 is reached exactly as SURVEY F3 describes:
by overriding the class attributes rows/cols/action_dim/state_dim.
"""
import sys
sys.dont_write_bytecode = True
import os
import io
import hashlib
import struct
import logging
import contextlib
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
sys.path.insert(0, os.path.dirname(HERE))           # tests/ -> fixture_eval
from fixture_eval import FixtureModel                 # noqa: E402

# ---- neutralise the reference's import-time file side effects ------------------
_real_makedirs = os.makedirs


def _makedirs(path, *a, **k):
    if os.path.abspath(str(path)).startswith(REF):
        return None
    return _real_makedirs(path, *a, **k)


os.makedirs = _makedirs
logging.FileHandler = lambda *a, **k: logging.NullHandler()
sys.path.insert(0, REF)
import games as ref_games          # noqa: E402
import ai as ref_ai                # noqa: E402
import utils as ref_utils          # noqa: E402
import train as ref_train          # noqa: E402
from replay_buffer import ReplayBuffer  # noqa: E402
from ai.nn import Net as RefNet    # noqa: E402

Node, MCTS = ref_ai.Node, ref_ai.MCTS
TTT, C4, GMK = ref_games.TicTacToe, ref_games.Connect4, ref_games.Gomoku


# ---- oracle-side shim: canonical form for 3-plane games -----------
def _canon3(board, player):
    if player == 0:
        return board
    out = np.empty_like(board)
    out[0], out[1], out[2] = board[1], board[0], board[2]
    return out


for _G in (TTT, C4):
    _G.feature_dim = 3
    _G.get_canonical_board = staticmethod(_canon3)


def set_gomoku(n):
    GMK.rows = GMK.cols = n
    GMK.action_dim = GMK.state_dim = n * n


def cells_of(Game, board):
    """int8 [R*C]: 0 empty, 1 player-0 stone, 2 player-1 stone."""
    return (board[0] + 2 * board[1]).astype(np.int8).reshape(-1)


def cell_idx(Game, mv):
    return mv[0] * Game.cols + mv[1]


def new_board(Game):
    return Game().board


# =================================================================================
# 1. rules
# =================================================================================
def gen_rules_ttt():
    """Exhaustive non-terminal reachable TicTacToe states (4520 = 5478 - 958 terminal) through the reference statics."""
    start = new_board(TTT)
    seen = {}
    order = []
    stack = [(start, 0, 0)]
    while stack:
        board, player, mc = stack.pop()
        key = cells_of(TTT, board).tobytes()
        if key in seen:
            continue
        seen[key] = len(order)
        valid = TTT.get_valid_moves(board)
        wins = np.full(9, -2, np.int8)
        for mv in valid:
            b2 = board.copy()
            nxt = TTT.make_move(b2, player, mv)
            assert nxt == 1 - player
            w = TTT.check_winner(b2, player, mv)
            wins[cell_idx(TTT, mv)] = w
            # undo restores the board bit-exactly
            b3 = b2.copy()
            TTT.undo_move(b3, nxt, mv)
            assert np.array_equal(b3[:2], board[:2])
            if w == -1 and mc + 1 < 9:
                stack.append((b2, nxt, mc + 1))
        vm = np.full(9, -1, np.int8)
        vm[:len(valid)] = [cell_idx(TTT, m) for m in valid]
        order.append((cells_of(TTT, board), player, vm, wins, board[2, 0, 0]))
    assert len(order) == 4520, len(order)   # 5478 reachable positions minus 958 terminal ones
    return dict(
        cells=np.stack([o[0] for o in order]),
        player=np.array([o[1] for o in order], np.int8),
        valid=np.stack([o[2] for o in order]),
        winner_after=np.stack([o[3] for o in order]),
        plane2=np.array([o[4] for o in order], np.float32),
    )


def gen_playouts(Game, n_playouts, seed):
    """Seeded uniform-random playouts; per ply: valid list (reference order), action, winner."""
    rng = np.random.RandomState(seed)
    acts, valid_flat, valid_off, winners, game_off, final = [], [], [0], [], [0], []
    for _ in range(n_playouts):
        g = Game()
        board, player, mc = g.board, 0, 0
        while True:
            valid = Game.get_valid_moves(board)
            valid_flat.extend(cell_idx(Game, m) for m in valid)
            valid_off.append(len(valid_flat))
            mv = valid[rng.randint(len(valid))]
            player_before = player
            player = Game.make_move(board, player, mv)
            mc += 1
            w = Game.check_winner(board, player_before, mv)
            acts.append(cell_idx(Game, mv))
            winners.append(w)
            if w != -1 or mc == Game.state_dim:
                break
        game_off.append(len(acts))
        final.append(cells_of(Game, board))
    return dict(
        actions=np.array(acts, np.int16), valid_flat=np.array(valid_flat, np.int16),
        valid_off=np.array(valid_off, np.int32), winners=np.array(winners, np.int8),
        game_off=np.array(game_off, np.int32), final_cells=np.stack(final),
    )


def gen_random_boards(Game, n_boards, n_queries, seed):
    """Arbitrary (not necessarily reachable) boards: valid-move lists + check_winner queries
    on arbitrary (player, cell) - including cells that do not hold the player's stone,
    because the reference's run counter starts at 1 without testing the origin cell."""
    rng = np.random.RandomState(seed)
    R, C = Game.rows, Game.cols
    planes = 2 if Game is GMK else 3
    cells_all, valid_flat, valid_off, q_all = [], [], [0], []
    for b in range(n_boards):
        dens = rng.choice([0.0, 0.02, 0.1, 0.3, 0.6, 0.9, 1.0])
        u = rng.rand(R * C)
        cells = np.where(u < dens / 2, 1, np.where(u < dens, 2, 0)).astype(np.int8)
        if Game is C4:   # gravity-consistent columns (get_drop_row semantics are column scans)
            grid = np.zeros((R, C), np.int8)
            for c in range(C):
                hgt = rng.randint(0, R + 1)
                col = rng.randint(1, 3, size=hgt)
                grid[R - hgt:, c] = col
            cells = grid.reshape(-1)
        board = np.zeros((planes, R, C), np.float32)
        board[0] = (cells == 1).reshape(R, C)
        board[1] = (cells == 2).reshape(R, C)
        valid = Game.get_valid_moves(board)
        valid_flat.extend(cell_idx(Game, m) for m in valid)
        valid_off.append(len(valid_flat))
        cells_all.append(cells)
        for _ in range(n_queries):
            p = rng.randint(2)
            r, c = rng.randint(R), rng.randint(C)
            q_all.append((b, p, r, c, Game.check_winner(board, p, (r, c))))
    return dict(cells=np.stack(cells_all), valid_flat=np.array(valid_flat, np.int16),
                valid_off=np.array(valid_off, np.int32), queries=np.array(q_all, np.int16))


# =================================================================================
# 2. search
# =================================================================================
def tree_digest(root, Game):
    """sha256 over the whole tree, DFS pre-order, children in list order."""
    h = hashlib.sha256()
    n_nodes = 0
    stack = [(root, 0)]
    while stack:
        node, depth = stack.pop()
        a = -1 if node.prevAction is None else cell_idx(Game, node.prevAction)
        h.update(struct.pack("<iiqdd", depth, a, int(node.visit), float(node.value), float(node.prior)))
        n_nodes += 1
        for ch in reversed(node.children):
            stack.append((ch, depth + 1))
    return h.hexdigest(), n_nodes


def random_position(Game, plies, rng):
    """Play `plies` uniformly random legal moves from the empty board (retry on early end)."""
    while True:
        g = Game()
        board, player, acts = g.board, 0, []
        ok = True
        for _ in range(plies):
            valid = Game.get_valid_moves(board)
            mv = valid[rng.randint(len(valid))]
            pb = player
            player = Game.make_move(board, player, mv)
            acts.append(cell_idx(Game, mv))
            if Game.check_winner(board, pb, mv) != -1 or len(acts) == Game.state_dim:
                ok = False
                break
        if ok:
            return board, player, acts


def gen_search_cases():
    out = {}
    meta = []
    rng = np.random.RandomState(1234)
    specs = []
    # (game name, Game, size, plies, n_sims, dirichlet, variant)
    for plies, n, dirichlet, variant in [(0, 25, True, "hash"), (1, 200, True, "hash"), (6, 200, True, "hash"),
                                         (12, 800, True, "hash"), (12, 200, False, "hash"),
                                         (5, 200, True, "uniform"), (20, 400, True, "uniform"),
                                         (30, 800, True, "hash")]:
        specs.append(("gomoku", 7, plies, n, dirichlet, variant))
    for plies, n, dirichlet, variant in [(0, 25, True, "hash"), (1, 200, True, "hash"), (10, 800, True, "hash"),
                                         (25, 800, True, "hash"), (25, 300, False, "hash"),
                                         (16, 400, True, "uniform"), (60, 800, True, "hash"),
                                         (120, 400, True, "hash")]:
        specs.append(("gomoku", 15, plies, n, dirichlet, variant))
    for plies, n, dirichlet, variant in [(0, 25, True, "hash"), (3, 200, True, "hash"), (5, 800, True, "uniform"),
                                         (6, 100, False, "hash")]:
        specs.append(("tictactoe", 3, plies, n, dirichlet, variant))
    for plies, n, dirichlet, variant in [(0, 200, True, "hash"), (8, 200, True, "hash"), (20, 800, True, "hash"),
                                         (14, 400, True, "uniform"), (30, 200, False, "hash")]:
        specs.append(("connect4", 0, plies, n, dirichlet, variant))

    for ci, (gname, size, plies, n_sims, dirichlet, variant) in enumerate(specs):
        Game = {"gomoku": GMK, "tictactoe": TTT, "connect4": C4}[gname]
        if gname == "gomoku":
            set_gomoku(size)
        board, player, acts = random_position(Game, plies, rng)
        noise_log = []
        orig_dir = np.random.dirichlet

        def rec_dir(alpha, size=None):
            x = orig_dir(alpha, size)
            noise_log.append(np.array(x, np.float64))
            return x
        np.random.dirichlet = rec_dir
        np.random.seed(1000 + ci)
        MCTS.cache.clear()
        MCTS.matched = 0
        MCTS.mcts_count = 0
        model = FixtureModel(Game.action_dim, variant)
        root = Node(None, None, player, len(acts))
        before = board.copy()
        with torch.no_grad():
            MCTS.mcts(model, board, root, Game, n_sims, dirichlet)
        np.random.dirichlet = orig_dir
        assert np.array_equal(before, board)          # board restored on exit (mcts.py contract)
        digest, n_nodes = tree_digest(root, Game)
        k = f"c{ci}_"
        out[k + "actions"] = np.array(acts, np.int16)
        out[k + "noise"] = noise_log[0] if noise_log else np.zeros(0)
        assert len(noise_log) == (1 if dirichlet else 0)
        out[k + "child_cell"] = np.array([cell_idx(Game, c.prevAction) for c in root.children], np.int16)
        out[k + "child_visit"] = np.array([c.visit for c in root.children], np.int64)
        out[k + "child_value"] = np.array([float(c.value) for c in root.children], np.float64)
        out[k + "child_prior"] = np.array([float(c.prior) for c in root.children], np.float64)
        out[k + "pi"] = ref_utils.get_probablity_distribution_of_children(root, Game).astype(np.float64)
        meta.append(dict(case=ci, game=gname, size=size, plies=plies, n_sims=n_sims, dirichlet=bool(dirichlet),
                         variant=variant, player=int(player), root_visit=int(root.visit),
                         root_value=float(root.value), digest=digest, n_nodes=n_nodes,
                         matched=int(MCTS.matched), mcts_count=int(MCTS.mcts_count), evals=int(model.calls),
                         prior_is_f64=bool(isinstance(root.children[0].prior, np.float64))))
        print("search case", meta[-1])
    import json
    out["meta_json"] = np.frombuffer(json.dumps(meta).encode(), np.uint8)
    return out


# =================================================================================
# 3. whole games (+ the caller contract train.save_data_to_buffer)
# =================================================================================
class RngRecorder:
    """Wraps the three global np.random entry points the hot path uses
    (utils.py:24 dirichlet, node.py:93 choice, mcts.py:73 randint) and logs the draws."""

    def __init__(self):
        self.noise, self.uniforms, self.randints = [], [], []

    def __enter__(self):
        self.o = (np.random.dirichlet, np.random.choice, np.random.randint)
        rec = self

        def dirichlet(alpha, size=None):
            x = rec.o[0](alpha, size)
            rec.noise.append(np.array(x, np.float64))
            return x

        def choice(a, size=None, replace=True, p=None):
            st = np.random.get_state()
            u = np.random.random_sample()           # legacy choice(p=...) consumes exactly one double
            np.random.set_state(st)
            r = rec.o[1](a, size=size, replace=replace, p=p)
            rec.uniforms.append(u)
            return r

        def randint(low, high=None, size=None, dtype=int):
            r = rec.o[2](low, high, size, dtype)
            rec.randints.append((int(low), int(r)))
            return r
        np.random.dirichlet, np.random.choice, np.random.randint = dirichlet, choice, randint
        return self

    def __exit__(self, *a):
        np.random.dirichlet, np.random.choice, np.random.randint = self.o


def buffer_digest(buf):
    h = hashlib.sha256()
    for state, pi, z in buf.buffer:
        h.update(np.ascontiguousarray(state, np.float32).tobytes())
        h.update(np.ascontiguousarray(pi, np.float64).tobytes())
        h.update(struct.pack("<d", float(z[0])))
    return h.hexdigest()


def gen_games():
    out, meta = {}, []
    import json
    specs = [("gomoku", 7, 100, "hash", 7), ("gomoku", 7, 60, "uniform", 8), ("gomoku", 15, 800, "hash", 0),
             ("gomoku", 15, 200, "uniform", 3), ("tictactoe", 3, 25, "hash", 11), ("connect4", 0, 200, "hash", 12),
             ("tictactoe", 3, 25, None, 0), ("connect4", 0, 50, None, 5),
             ("gomoku", 7, 40, None, 9), ("gomoku", 15, 16, None, 4)]       # vanilla Gomoku: rollouts through get_valid_moves' set order
    for gi, (gname, size, n_sims, variant, seed) in enumerate(specs):
        Game = {"gomoku": GMK, "tictactoe": TTT, "connect4": C4}[gname]
        if gname == "gomoku":
            set_gomoku(size)
        MCTS.cache.clear()
        MCTS.matched = 0
        MCTS.mcts_count = 0
        model = FixtureModel(Game.action_dim, variant) if variant else None
        np.random.seed(seed)
        t0 = time.time()
        with RngRecorder() as rec, torch.no_grad(), contextlib.redirect_stdout(io.StringIO()):
            res = Game().self_play(model, n_sims)
        dt = time.time() - t0
        if gname == "gomoku":
            boards, actions, pis, qs, winner = res
            act_cells = np.array([cell_idx(Game, a) for a in actions[1:]], np.int16)
        else:
            boards, pis, winner = res
            qs = []
            # actions are not returned for TTT/C4: recover them from successive boards
            act_cells = []
            for t in range(len(boards)):
                nxt = boards[t + 1] if t + 1 < len(boards) else None
                if nxt is not None:
                    d = (nxt[0] + nxt[1]) - (boards[t][0] + boards[t][1])
                    act_cells.append(int(np.argmax(d.reshape(-1))))
            act_cells = np.array(act_cells, np.int16)      # last move unknown from outputs (length T-1)
        k = f"g{gi}_"
        out[k + "actions"] = act_cells
        out[k + "pis"] = np.stack(pis).astype(np.float64)
        out[k + "qs"] = np.array([float(q) for q in qs], np.float64)
        out[k + "board_cells"] = np.stack([cells_of(Game, b) for b in boards])
        out[k + "noise"] = np.stack(rec.noise) if rec.noise else np.zeros((0, Game.action_dim))
        out[k + "uniforms"] = np.array(rec.uniforms, np.float64)
        out[k + "randints"] = np.array(rec.randints, np.int32).reshape(-1, 2)
        m = dict(game=gi, name=gname, size=size, n_sims=n_sims, variant=variant, seed=seed, winner=int(winner),
                 n_moves=len(boards), matched=int(MCTS.matched), mcts_count=int(MCTS.mcts_count),
                 evals=int(model.calls) if model else 0, ref_seconds=round(dt, 2))
        if gname == "gomoku":
            buf = ReplayBuffer(100000)
            reward = 0 if winner == -1 else 1
            ref_train.save_data_to_buffer(Game, buf, (boards, actions, pis, qs, winner, reward))
            m["buffer_len"] = buf.size()
            m["buffer_digest"] = buffer_digest(buf)
        meta.append(m)
        print("game", m)
    out["meta_json"] = np.frombuffer(json.dumps(meta).encode(), np.uint8)
    return out


# =================================================================================
# 3b. test.compete (test.py:60-105): two agents alternating by side, optional sampling for move_count < 20
# =================================================================================
def gen_compete():
    import importlib
    ref_test = importlib.import_module("test")          # /root/reference/test.py (sys.path[0] = REF shadows the stdlib package)
    assert os.path.abspath(ref_test.__file__).startswith(REF)
    import json
    out, meta = {}, []
    specs = [(7, "hash", "uniform", 40, 24, True, 21), (7, "uniform", "hash", 30, 30, False, 22), (15, "hash", "uniform", 60, 40, True, 23),
             (7, None, "hash", 30, 30, False, 24)]          # last one: vanilla MCTS as player 0 (main.py:76 shape)
    for ci, (size, v1, v2, it1, it2, sampling, seed) in enumerate(specs):
        set_gomoku(size)
        MCTS.cache.clear()
        MCTS.matched = 0
        MCTS.mcts_count = 0
        m1 = FixtureModel(GMK.action_dim, v1) if v1 else None
        m2 = FixtureModel(GMK.action_dim, v2) if v2 else None
        np.random.seed(seed)
        with RngRecorder() as rec, torch.no_grad(), contextlib.redirect_stdout(io.StringIO()):
            winner, board = ref_test.compete(GMK, m1, m2, it1, it2, sampling=sampling, display=False)
        k = f"c{ci}_"
        out[k + "final_cells"] = cells_of(GMK, board)
        out[k + "noise"] = np.stack(rec.noise) if rec.noise else np.zeros((0, GMK.action_dim))
        out[k + "uniforms"] = np.array(rec.uniforms, np.float64)
        out[k + "randints"] = np.array(rec.randints, np.int32).reshape(-1, 2)
        m = dict(case=ci, size=size, variant1=v1, variant2=v2, iter1=it1, iter2=it2, sampling=sampling, seed=seed,
                 winner=int(winner), n_moves=int((board[0] + board[1]).sum()), mcts_count=int(MCTS.mcts_count), matched=int(MCTS.matched))
        meta.append(m)
        print("compete", m)
    # test.compare (test.py:107-140): a series of compete games, the models swap sides at half time, MCTS.cache is NOT cleared
    # in between; recorded: the seed, the per-game winners (by wrapping compete) and the returned value
    cmeta = []
    cspecs = [(7, "hash", "uniform", 30, 20, 4, False, False, 31), (7, "uniform", "hash", 24, 24, 6, True, True, 32),
              (7, None, "hash", 20, 20, 2, False, False, 33)]
    for ki, (size, vb, vc, itb, itc, iters, sampling, early, seed) in enumerate(cspecs):
        set_gomoku(size)
        MCTS.cache.clear()
        MCTS.matched = 0
        MCTS.mcts_count = 0
        mb = FixtureModel(GMK.action_dim, vb) if vb else None
        mc_ = FixtureModel(GMK.action_dim, vc) if vc else None
        winners = []
        real_compete = ref_test.compete

        def spy(*a, **k):
            w, b = real_compete(*a, **k)
            winners.append(int(w))
            return w, b
        ref_test.compete = spy
        np.random.seed(seed)
        try:
            with torch.no_grad(), contextlib.redirect_stdout(io.StringIO()):
                value = ref_test.compare(GMK, mb, mc_, itb, itc, iters, sampling, early)
        finally:
            ref_test.compete = real_compete
        cm = dict(case=ki, size=size, best=vb, contender=vc, best_iter=itb, contender_iter=itc, iterations=iters, sampling=sampling,
                  early_stopping=early, seed=seed, winners=winners, value=float(value), mcts_count=int(MCTS.mcts_count), matched=int(MCTS.matched))
        cmeta.append(cm)
        print("compare", cm)
    out["compare_meta_json"] = np.frombuffer(json.dumps(cmeta).encode(), np.uint8)
    out["meta_json"] = np.frombuffer(json.dumps(meta).encode(), np.uint8)
    return out


# =================================================================================
# 4. network known-answer test (small config; weights committed as data)
# =================================================================================
def gen_nn_small():
    out = {}
    torch.manual_seed(0)
    cfgs = [dict(img_size=7, patch_size=5, embed_dim=32, action_dim=49, num_heads=4, depth=2, channels=2),
            dict(img_size=15, patch_size=3, embed_dim=32, action_dim=225, num_heads=2, depth=1, channels=2)]
    import json
    for ni, cfg in enumerate(cfgs):
        net = RefNet(dropout=0.1, **cfg).eval()
        rng = np.random.RandomState(ni)
        x = np.zeros((6, cfg["channels"], cfg["img_size"], cfg["img_size"]), np.float32)
        for b in range(6):
            u = rng.rand(cfg["img_size"], cfg["img_size"])
            x[b, 0] = u < 0.15 * b / 5
            x[b, 1] = (u >= 0.15 * b / 5) & (u < 0.3 * b / 5)
        with torch.no_grad():
            logits, v = net(torch.from_numpy(x))
        k = f"n{ni}_"
        for name, t in net.state_dict().items():
            out[k + "sd_" + name] = t.numpy()
        out[k + "x"] = x
        out[k + "logits"] = logits.numpy()
        out[k + "value"] = v.numpy()
        out[k + "cfg_json"] = np.frombuffer(json.dumps(cfg).encode(), np.uint8)
    # full training config (main.py:134 at 15x15): weights are NOT committed (13.65 MB); the key set,
    # shapes and a seed-0 output KAT are, so the build's own initialiser can be checked against it.
    torch.manual_seed(0)
    net = RefNet(15, patch_size=5, embed_dim=512, action_dim=225, num_heads=8, depth=1, channels=2, dropout=0.1).eval()
    out["full_keys_json"] = np.frombuffer(json.dumps({k: list(v.shape) for k, v in net.state_dict().items()}).encode(), np.uint8)
    x = np.zeros((4, 2, 15, 15), np.float32)
    x[1, 0, 7, 7] = 1
    x[2, 0, 7, 7] = 1; x[2, 1, 6, 8] = 1
    x[3, 1, 7, 7] = 1; x[3, 0, 6, 8] = 1; x[3, 0, 0, 0] = 1
    with torch.no_grad():
        logits, v = net(torch.from_numpy(x))
    out["full_x"] = x
    out["full_logits"] = logits.numpy()
    out["full_value"] = v.numpy()
    out["full_param_sums"] = np.array([float(t.double().sum()) for t in net.state_dict().values()], np.float64)
    return out


def gen_nn_depth2():
    """main.py:186-188's network (the compare mode builds Net(rows, patch_size=5, embed_dim=256, num_heads=8, depth=2)) at 15x15 under
    torch.manual_seed(0): outputs on eight boards + the parameter sums (the weights themselves are not committed: the build's
    initialiser reproduces them from the seed, checked against the sums)."""
    import json
    out = {}
    torch.manual_seed(0)
    cfg = dict(img_size=15, patch_size=5, embed_dim=256, action_dim=225, num_heads=8, depth=2, channels=2)
    net = RefNet(dropout=0.1, **cfg).eval()
    rng = np.random.RandomState(11)
    x = np.zeros((8, 2, 15, 15), np.float32)
    for b in range(8):
        k = 6 * b
        cells = rng.choice(225, size=k, replace=False)
        x[b, 0].reshape(-1)[cells[: k // 2]] = 1
        x[b, 1].reshape(-1)[cells[k // 2:]] = 1
    with torch.no_grad():
        logits, v = net(torch.from_numpy(x))
    out["x"], out["logits"], out["value"] = x, logits.numpy(), v.numpy()
    out["cfg_json"] = np.frombuffer(json.dumps(cfg).encode(), np.uint8)
    out["keys_json"] = np.frombuffer(json.dumps({k: list(v_.shape) for k, v_ in net.state_dict().items()}).encode(), np.uint8)
    out["param_sums"] = np.array([float(t.double().sum()) for t in net.state_dict().values()], np.float64)
    return out


# =================================================================================
# 5. train step (train.py:85-123): loss, L2 quirk, Adam - small net, dropout 0 (deterministic)
# =================================================================================
def gen_train(dropout=0.0, torch_seed_for_masks=None):
    """dropout > 0: the reference's actual training configuration (main.py:134 dropout=0.1, model.train() at train.py:92);
    torch is re-seeded right before train.train so that the dropout masks are a function of `torch_seed_for_masks` alone."""
    import json
    out = {}
    torch.manual_seed(3)
    cfg = dict(img_size=7, patch_size=5, embed_dim=32, action_dim=49, num_heads=4, depth=2, channels=2)
    net = RefNet(dropout=dropout, **cfg)
    rng = np.random.RandomState(5)
    B = 24
    states = (rng.rand(B, 2, 7, 7) < 0.2).astype(np.float32)
    states[:, 1] *= 1 - states[:, 0]
    pis = rng.dirichlet([0.5] * 49, size=B)                       # float64, as the buffer holds them
    zs = rng.choice([-1.0, 0.0, 1.0], size=B)
    buf = ReplayBuffer(1000)
    for b in range(B):
        buf.add(states[b], pis[b], [float(zs[b])])
    for name, t in net.state_dict().items():
        out["init_" + name] = t.numpy().copy()
    order = []
    orig_choice = np.random.choice

    def rec_choice(a, size=None, replace=True, p=None):
        r = orig_choice(a, size=size, replace=replace, p=p)
        order.append(np.array(r))
        return r
    np.random.choice = rec_choice
    np.random.seed(11)
    if torch_seed_for_masks is not None:
        torch.manual_seed(torch_seed_for_masks)
        out["torch_seed_for_masks"] = np.array(torch_seed_for_masks)
        out["torch_threads"] = np.array(torch.get_num_threads())
    out["dropout"] = np.array(dropout)
    with contextlib.redirect_stdout(io.StringIO()):
        losses = ref_train.train(net, B, buf, 3, 0.00025, "cpu")
    np.random.choice = orig_choice
    for name, t in net.state_dict().items():
        out["final_" + name] = t.detach().numpy().copy()
    out["states"], out["pis"], out["zs"] = states, pis, zs
    out["batch_order"] = np.stack(order)
    out["losses_after_3"] = np.array(losses, np.float64)          # (loss, policy_loss, value_loss, l2) of iteration 3
    out["cfg_json"] = np.frombuffer(json.dumps(cfg).encode(), np.uint8)
    print("train losses", losses)
    return out


def main():
    which = sys.argv[1:] or ["rules", "search", "games", "compete", "nn", "nn_depth2", "train", "train_dropout"]
    print("python", sys.version.split()[0], "numpy", np.__version__, "torch", torch.__version__,
          "cpus", os.cpu_count(), "torch threads", torch.get_num_threads())
    if "rules" in which:
        np.savez_compressed(os.path.join(HERE, "rules_ttt.npz"), **gen_rules_ttt())
        np.savez_compressed(os.path.join(HERE, "rules_ttt_rand.npz"), **gen_random_boards(TTT, 60, 30, 5))
        np.savez_compressed(os.path.join(HERE, "rules_c4.npz"), **gen_playouts(C4, 60, 1),
                            **{"rb_" + k: v for k, v in gen_random_boards(C4, 80, 40, 2).items()})
        set_gomoku(7)
        np.savez_compressed(os.path.join(HERE, "rules_gomoku7.npz"), **gen_playouts(GMK, 40, 3),
                            **{"rb_" + k: v for k, v in gen_random_boards(GMK, 80, 40, 4).items()})
        set_gomoku(15)
        np.savez_compressed(os.path.join(HERE, "rules_gomoku15.npz"), **gen_playouts(GMK, 12, 5),
                            **{"rb_" + k: v for k, v in gen_random_boards(GMK, 60, 60, 6).items()})
    if "search" in which:
        np.savez_compressed(os.path.join(HERE, "search.npz"), **gen_search_cases())
    if "games" in which:
        np.savez_compressed(os.path.join(HERE, "games.npz"), **gen_games())
    if "compete" in which:
        np.savez_compressed(os.path.join(HERE, "compete.npz"), **gen_compete())
    if "nn" in which:
        np.savez_compressed(os.path.join(HERE, "nn_small.npz"), **gen_nn_small())
    if "nn_depth2" in which:
        np.savez_compressed(os.path.join(HERE, "nn_depth2.npz"), **gen_nn_depth2())
    if "train" in which:
        np.savez_compressed(os.path.join(HERE, "train_small.npz"), **gen_train())
    if "train_dropout" in which:
        np.savez_compressed(os.path.join(HERE, "train_dropout.npz"), **gen_train(dropout=0.1, torch_seed_for_masks=21))
    assert not os.path.exists(os.path.join(REF, "logs")), "reference tree was written to!"
    assert not os.path.exists(os.path.join(REF, "__pycache__"))


if __name__ == "__main__":
    main()
