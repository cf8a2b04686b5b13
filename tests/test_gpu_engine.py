"""GPU parity tests: the HIP engine (through the C ABI, via the azk ctypes binding) against
 (a) the reference's golden vectors (tests/golden, produced by running the reference) and
 (b) the C oracle on the same seeded inputs.
Bit-exact: legal-move lists/order, terminal detection, every node of every tree (N, W, P, order),
pi, q, chosen actions, winners.  The oracle is used here only as the checker."""
import hashlib
import struct

import numpy as np
import pytest
import torch

from conftest import golden_meta, load_golden
from fixture_eval import fixture_logits_value

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def azk():
    import azk as m
    m.lib()
    return m


@pytest.fixture(scope="module")
def ao():
    from oracle import az_oracle
    return az_oracle


def dev():
    return torch.device("cuda", 0)


def boards_from_cells(cells_list, planes, rows, cols):
    out = np.zeros((len(cells_list), planes, rows, cols), np.float32)
    for i, c in enumerate(cells_list):
        c = np.asarray(c).reshape(rows, cols)
        out[i, 0] = c == 1
        out[i, 1] = c == 2
    return out


# ---------------------------------------------------------------------------------------------------
# board rules vs golden
# ---------------------------------------------------------------------------------------------------
GAMES = [("tictactoe", None, 3, 3, 3, "rules_ttt_rand.npz", ""), ("connect4", None, 3, 6, 7, "rules_c4.npz", "rb_"),
         ("gomoku", 7, 2, 7, 7, "rules_gomoku7.npz", "rb_"), ("gomoku", 15, 2, 15, 15, "rules_gomoku15.npz", "rb_")]


@pytest.mark.parametrize("name,size,planes,rows,cols,file,prefix", GAMES)
def test_rules_random_boards(azk, name, size, planes, rows, cols, file, prefix):
    z = load_golden(file)
    cells = z[prefix + "cells"]
    boards = torch.from_numpy(boards_from_cells(cells, planes, rows, cols)).to(dev())
    moves, counts = azk.rules_legal_moves(name, boards, size)
    moves, counts = moves.cpu().numpy(), counts.cpu().numpy()
    A = 7 if name == "connect4" else rows * cols
    mask = azk.rules_legal_mask(name, boards, A, size).cpu().numpy()
    for bi in range(len(cells)):
        want = z[prefix + "valid_flat"][z[prefix + "valid_off"][bi]:z[prefix + "valid_off"][bi + 1]].tolist()
        assert moves[bi, :counts[bi]].tolist() == want, (name, bi)
        wm = np.zeros(A, np.uint8)
        for c in want:
            wm[c % cols if name == "connect4" else c] = 1
        assert np.array_equal(mask[bi], wm)
    q = z[prefix + "queries"]
    qb = boards[torch.from_numpy(q[:, 0].astype(np.int64)).to(dev())].contiguous()
    players = torch.from_numpy(q[:, 1].astype(np.int32)).to(dev())
    cellq = torch.from_numpy((q[:, 2].astype(np.int32) * cols + q[:, 3].astype(np.int32))).to(dev())
    w = azk.rules_check_winner(name, qb, players, cellq, size).cpu().numpy()
    assert w.tolist() == q[:, 4].astype(np.int32).tolist()


def test_rules_tictactoe_exhaustive(azk):
    z = load_golden("rules_ttt.npz")
    cells = z["cells"]
    boards_np = boards_from_cells(cells, 3, 3, 3)
    boards_np[:, 2] = z["plane2"][:, None, None]
    boards = torch.from_numpy(boards_np).to(dev())
    moves, counts = azk.rules_legal_moves("tictactoe", boards)
    moves, counts = moves.cpu().numpy(), counts.cpu().numpy()
    for i in range(len(cells)):
        assert moves[i, :counts[i]].tolist() == [int(v) for v in z["valid"][i] if v >= 0]
    # every legal (state, move): apply, winner, undo
    idx, mv = np.nonzero(z["winner_after"] != -2)
    b = boards[torch.from_numpy(idx).to(dev())].contiguous()
    players = torch.from_numpy(z["player"][idx].astype(np.int32)).to(dev())
    cellt = torch.from_numpy(mv.astype(np.int32)).to(dev())
    before = b.clone()
    nxt = azk.rules_apply_move("tictactoe", b, players, cellt)
    assert torch.equal(nxt, 1 - players)
    assert torch.equal(b[:, 2, 0, 0], (1 - players).float())
    w = azk.rules_check_winner("tictactoe", b, players, cellt).cpu().numpy()
    assert w.tolist() == z["winner_after"][idx, mv].astype(np.int32).tolist()
    azk.rules_undo_move("tictactoe", b, nxt, cellt)
    assert torch.equal(b[:, :2], before[:, :2])
    assert torch.equal(b[:, 2, 0, 0], players.float())


@pytest.mark.parametrize("name,size,file", [("connect4", None, "rules_c4.npz"), ("gomoku", 7, "rules_gomoku7.npz"),
                                            ("gomoku", 15, "rules_gomoku15.npz")])
def test_rules_playouts(azk, ao, name, size, file):
    """Replay the reference's playouts with the apply-move kernel; legal lists and winners at every ply."""
    z = load_golden(file)
    og = ao.OracleGame(name, size)
    n_games = len(z["game_off"]) - 1
    boards = torch.zeros((n_games, og.planes, og.rows, og.cols), dtype=torch.float32, device=dev())
    lens = np.diff(z["game_off"])
    for t in range(int(lens.max())):
        live = np.nonzero(lens > t)[0]
        ply = z["game_off"][live] + t
        sub = boards[torch.from_numpy(live).to(dev())].contiguous()
        moves, counts = azk.rules_legal_moves(name, sub, size)
        moves, counts = moves.cpu().numpy(), counts.cpu().numpy()
        for j, p in enumerate(ply):
            want = z["valid_flat"][z["valid_off"][p]:z["valid_off"][p + 1]].tolist()
            assert moves[j, :counts[j]].tolist() == want, (name, live[j], t)
        players = torch.full((len(live),), t & 1, dtype=torch.int32, device=dev())
        cellt = torch.from_numpy(z["actions"][ply].astype(np.int32)).to(dev())
        nxt = azk.rules_apply_move(name, sub, players, cellt, size)
        assert torch.equal(nxt, 1 - players)
        w = azk.rules_check_winner(name, sub, players, cellt, size).cpu().numpy()
        assert w.tolist() == z["winners"][ply].astype(np.int32).tolist()
        boards[torch.from_numpy(live).to(dev())] = sub
    final = (boards[:, 0] + 2 * boards[:, 1]).to(torch.int8).reshape(n_games, -1).cpu().numpy()
    assert np.array_equal(final, z["final_cells"])


def test_rules_invalid_move_and_canonical(azk):
    b = torch.zeros((2, 2, 7, 7), dtype=torch.float32, device=dev())
    p0 = torch.tensor([0, 0], dtype=torch.int32, device=dev())
    c = torch.tensor([10, 11], dtype=torch.int32, device=dev())
    assert azk.rules_apply_move("gomoku", b, p0, c, 7).tolist() == [1, 1]
    before = b.clone()
    p1 = torch.tensor([1, 1], dtype=torch.int32, device=dev())
    assert azk.rules_apply_move("gomoku", b, p1, c, 7).tolist() == [1, 1]      # occupied: unchanged player, board untouched
    assert torch.equal(b, before)
    can = azk.rules_canonical("gomoku", b, torch.tensor([0, 1], dtype=torch.int32, device=dev()), 7)
    assert torch.equal(can[0], b[0]) and torch.equal(can[1, 0], b[1, 1]) and torch.equal(can[1, 1], b[1, 0])
    b4 = torch.zeros((1, 3, 6, 7), dtype=torch.float32, device=dev())
    assert azk.rules_apply_move("connect4", b4, torch.tensor([0], dtype=torch.int32, device=dev()), torch.tensor([38], dtype=torch.int32, device=dev())).tolist() == [1]
    assert azk.rules_apply_move("connect4", b4, torch.tensor([1], dtype=torch.int32, device=dev()), torch.tensor([38], dtype=torch.int32, device=dev())).tolist() == [0]
    assert b4[0, 0, 5, 3] == 1 and b4[0, 1, 5, 3] == 1                                 # connect4.py:56-63: no occupancy check


# ---------------------------------------------------------------------------------------------------
# numerics: the softmax the engine applies == the oracle's deterministic softmax, bit for bit
# ---------------------------------------------------------------------------------------------------
def test_softmax_rows_bit_exact(azk, ao):
    rng = np.random.RandomState(0)
    for A in (7, 9, 49, 225, 361):
        l = rng.uniform(-3, 3, (64, A)).astype(np.float32)
        l[0] = 0
        l[1, 0] = 30.0
        l[2] = -80.0
        got = azk.softmax_rows(torch.from_numpy(l).to(dev())).cpu().numpy()
        for i in range(len(l)):
            assert got[i].tobytes() == ao.softmax_det(l[i]).tobytes(), (A, i)


# ---------------------------------------------------------------------------------------------------
# search: whole trees
# ---------------------------------------------------------------------------------------------------
_SZ = load_golden("search.npz")
_SMETA = golden_meta(_SZ)


def gpu_evaluator(A, variant):
    def ev(x):
        return fixture_logits_value(x, A, variant)
    return ev


def digest(e):
    h = hashlib.sha256()
    for d, c, n, w, p in zip(e["depth"], e["cell"], e["visit"], e["value"], e["prior"]):
        h.update(struct.pack("<iiqdd", int(d), int(c), int(n), float(w), float(p)))
    return h.hexdigest(), len(e["depth"])


def oracle_tree(ao, m, k, softmax):
    game = ao.OracleGame(m["game"], m["size"] or None)
    b = game.new_board()
    player = 0
    for cell in _SZ[k + "actions"]:
        player = game.make_move(b, player, game.rc(int(cell)))
    tree = ao.OracleTree(game)
    tree.reset(player, len(_SZ[k + "actions"]))

    def ev(canon):
        logits, v = fixture_logits_value(torch.from_numpy(np.ascontiguousarray(canon))[None], game.action_dim, m["variant"])
        return softmax(logits[0].numpy()), float(v[0])
    cnt = ao.Counters()
    ao.mcts(game, tree, b, m["n_sims"], ev, _SZ[k + "noise"] if m["dirichlet"] else None, None, None, cnt)
    cells = (b[0] + 2 * b[1]).astype(np.int8).reshape(-1)
    return game, tree, cells, player, cnt


@pytest.mark.parametrize("m", _SMETA, ids=[f"{m['case']}-{m['game']}{m['size']}-p{m['plies']}-n{m['n_sims']}-{m['variant']}" for m in _SMETA])
def test_search_tree_vs_oracle_and_golden(azk, ao, m):
    k = f"c{m['case']}_"
    game, tree, cells, player, cnt = oracle_tree(ao, m, k, ao.softmax_det)
    G = 3                                   # same position in several slots: every slot must agree
    eng = azk.Engine(m["game"], G, m["n_sims"], size=m["size"] or None)
    eng.set_positions(np.tile(cells, (G, 1)), [player] * G, [len(_SZ[k + "actions"])] * G)
    noise = None
    if m["dirichlet"]:
        noise = torch.from_numpy(np.tile(_SZ[k + "noise"], (G, 1))).to(dev())
    eng.search(gpu_evaluator(game.action_dim, m["variant"]), m["n_sims"], noise)
    eng.check_error()
    want = digest(tree.export())
    for g in range(G):
        assert digest(eng.export_tree(g)) == want, g
    ch = eng.root_children(1)
    # against the reference's own numbers: order, visits and W bit-exact; priors within float32 ulps
    assert ch["cell"].tolist() == _SZ[k + "child_cell"].tolist()
    assert ch["visit"].tolist() == _SZ[k + "child_visit"].tolist()
    assert ch["value"].tobytes() == _SZ[k + "child_value"].tobytes()
    np.testing.assert_allclose(ch["prior"], _SZ[k + "child_prior"], rtol=1e-6, atol=0)
    pi, q, rv = eng.root_stats()
    assert pi[0].cpu().numpy().tobytes() == _SZ[k + "pi"].tobytes()
    assert rv.tolist() == [m["root_visit"]] * G
    assert q[2].item() == m["root_value"] / m["root_visit"]
    c = eng.counters()
    assert c["sims"] == G * m["n_sims"]
    assert c["edges_scanned"] == G * cnt.edges_scanned and c["trace_nodes"] == G * cnt.trace_nodes
    assert c["edges_created"] == G * cnt.edges_created and c["terminal_sims"] == G * cnt.terminal_sims
    assert c["leaves_evaluated"] == G * cnt.expansions
    # the caller's positions are untouched by a search (mcts.py restores the board)
    cells_after, tm, mc = eng.get_positions()
    assert np.array_equal(cells_after[0], cells) and tm[0] == player
    eng.close()


def test_split_calls_equal_fused_calls(azk):
    """azk_step_select + azk_step_expand_backup (SURVEY 8(b) surface) == fused azk_step."""
    m = next(x for x in _SMETA if x["game"] == "gomoku" and x["size"] == 7 and x["n_sims"] == 200 and x["dirichlet"])
    k = f"c{m['case']}_"
    A = 49
    ev = gpu_evaluator(A, m["variant"])
    noise = torch.from_numpy(np.tile(_SZ[k + "noise"], (2, 1))).to(dev())
    trees = []
    for fused in (True, False):
        eng = azk.Engine("gomoku", 2, 200, size=7)
        eng.reset_games()
        if fused:
            eng.search(ev, 200, noise)
        else:
            eng.begin_search(noise)
            for _ in range(200):
                eng.step_select()
                n = int(eng.n_leaf.item())
                if n:
                    lg, v = ev(eng.leaf_boards[:n])
                    eng.step_expand_backup(lg.contiguous(), v.contiguous())
        trees.append(digest(eng.export_tree(0)))
        eng.close()
    assert trees[0] == trees[1]


# ---------------------------------------------------------------------------------------------------
# whole games
# ---------------------------------------------------------------------------------------------------
_GZ = load_golden("games.npz")
_GMETA = [m for m in golden_meta(_GZ) if m["variant"]]


@pytest.mark.parametrize("m", _GMETA, ids=[f"g{m['game']}-{m['name']}{m['size']}-n{m['n_sims']}-{m['variant']}" for m in _GMETA])
def test_self_play_vs_oracle_and_golden(azk, ao, m):
    """Engine self-play fed the reference's recorded Dirichlet draws and choice() uniforms.
    vs oracle (deterministic softmax on both sides): everything bit-exact.
    vs golden (reference used numpy's softmax): actions, winner, pi bit-exact on these games."""
    from selfplay import self_play_batch
    k = f"g{m['game']}_"
    noise, uniforms = _GZ[k + "noise"], _GZ[k + "uniforms"]
    size = m["size"] or None
    og = ao.OracleGame(m["name"], size)
    A = og.action_dim
    G = 2
    su = 8 if m["name"] == "gomoku" else 1 << 30

    def noise_fn(mv):
        return np.tile(noise[min(mv, len(noise) - 1)], (G, 1))

    def uniform_fn(mv):
        u = uniforms[mv] if mv < len(uniforms) else 0.5
        return np.full(G, u)

    res = self_play_batch(m["name"], gpu_evaluator(A, m["variant"]), G, m["n_sims"], size=size,
                          noise_fn=noise_fn, uniform_fn=uniform_fn)

    def ev(canon):
        logits, v = fixture_logits_value(torch.from_numpy(np.ascontiguousarray(canon))[None], A, m["variant"])
        return ao.softmax_det(logits[0].numpy()), float(v[0])
    out = ao.self_play(og, ev, m["n_sims"], noise_fn=lambda mv: noise[mv], uniform_fn=lambda mv: uniforms[mv])
    for g in range(G):
        r = res[g]
        assert r.winner == out["winner"] == m["winner"]
        assert r.cells == out["cells"].tolist()
        assert np.stack(r.pis).tobytes() == out["pis"].tobytes()
        assert np.array(r.qs).tobytes() == out["qs"].tobytes()
        assert all(np.array_equal(a, b) for a, b in zip(r.boards, out["boards"]))
        # reference
        assert len(r.boards) == m["n_moves"]
        assert np.stack(r.pis).tobytes() == _GZ[k + "pis"].tobytes()
        ref_actions = _GZ[k + "actions"]
        assert r.cells[:len(ref_actions)] == ref_actions.tolist()
        got_cells = np.stack([(b[0] + 2 * b[1]).astype(np.int8).reshape(-1) for b in r.boards])
        assert np.array_equal(got_cells, _GZ[k + "board_cells"])
        if m["name"] == "gomoku":
            assert np.array(r.qs).tobytes() == _GZ[k + "qs"].tobytes()


def test_many_independent_games_match_oracle(azk, ao):
    """A ragged batch: 48 Gomoku-7 games with different noise / uniforms, finishing at different plies.
    Every game must equal the oracle's sequential run of that game."""
    from selfplay import self_play_batch
    G, n_sims, A = 48, 64, 49
    rng = np.random.RandomState(5)
    T = 49
    noise = rng.dirichlet([0.3] * A, size=(T, G))
    uniforms = rng.random_sample((T, G))
    res = self_play_batch("gomoku", gpu_evaluator(A, "hash"), G, n_sims, size=7,
                          noise_fn=lambda mv: noise[mv], uniform_fn=lambda mv: uniforms[mv])
    og = ao.OracleGame("gomoku", 7)

    def ev(canon):
        logits, v = fixture_logits_value(torch.from_numpy(np.ascontiguousarray(canon))[None], A, "hash")
        return ao.softmax_det(logits[0].numpy()), float(v[0])
    lengths = set()
    for g in range(G):
        out = ao.self_play(og, ev, n_sims, noise_fn=lambda mv: noise[mv, g], uniform_fn=lambda mv: uniforms[mv, g])
        r = res[g]
        assert r.winner == out["winner"], g
        assert r.cells == out["cells"].tolist(), g
        assert np.stack(r.pis).tobytes() == out["pis"].tobytes(), g
        assert np.array(r.qs).tobytes() == out["qs"].tobytes(), g
        lengths.add(len(r.cells))
    assert len(lengths) > 3      # the batch really was ragged


def test_engine_noise_generator_properties(azk):
    """The product RNG: rows are valid Dirichlet draws, depend only on (seed, global game, move),
    and are independent of how games are sharded."""
    e1 = azk.Engine("gomoku", 8, 4, size=15)
    e2 = azk.Engine("gomoku", 4, 4, size=15)
    n1, u1 = e1.gen_noise(7, 100, 3)
    n2, u2 = e2.gen_noise(7, 104, 3)
    assert torch.equal(n1[4:], n2) and torch.equal(u1[4:], u2)
    assert torch.allclose(n1.sum(1), torch.ones(8, dtype=torch.float64, device=dev()), atol=1e-12)
    assert (n1 >= 0).all() and (u1 >= 0).all() and (u1 < 1).all()
    n3, _ = e1.gen_noise(8, 100, 3)
    assert not torch.equal(n1, n3)
    # alpha = 0.03 concentrates mass.  numpy's Dirichlet([0.03]*225) has E[max] = 0.255 and on average
    # 28.65 entries above 1e-3 (4096 draws); the engine's generator must land on the same statistics.
    big, _ = azk.Engine("gomoku", 2048, 4, size=15).gen_noise(1, 0, 0)
    assert 0.235 < big.max(1).values.mean().item() < 0.275
    assert 27.0 < (big > 1e-3).sum(1).double().mean().item() < 30.3
    # Dirichlet marginal mean = 1/A
    assert abs(big.mean().item() - 1 / 225) < 1e-9
    assert abs(big[:, 0].mean().item() - 1 / 225) < 0.01


def test_arena_overflow_is_an_error_not_ub(azk):
    eng = azk.Engine("gomoku", 2, 50, size=7, arena_nodes=40)
    eng.reset_games()
    eng.search(gpu_evaluator(49, "hash"), 50, None)
    with pytest.raises(azk.AzkError):
        eng.check_error()


# ---------------------------------------------------------------------------------------------------
# eval cache (MCTS.cache): transparent - same trees, same games - and it does hit
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("case", [3, 10, 13, 18, 22])
def test_eval_cache_is_transparent_on_trees(azk, case):
    m = next(x for x in _SMETA if x["case"] == case)
    k = f"c{m['case']}_"
    from oracle import az_oracle as ao
    game, tree, cells, player, cnt = oracle_tree(ao, m, k, ao.softmax_det)
    eng = azk.Engine(m["game"], 2, m["n_sims"], size=m["size"] or None, cache_entries=4096)
    eng.set_positions(np.tile(cells, (2, 1)), [player] * 2, [len(_SZ[k + "actions"])] * 2)
    noise = torch.from_numpy(np.tile(_SZ[k + "noise"], (2, 1))).to(dev()) if m["dirichlet"] else None
    want = digest(tree.export())
    for rep in range(2):                    # second search of the same position: (almost) everything comes from the cache
        eng.reset_counters()
        eng.search(gpu_evaluator(game.action_dim, m["variant"]), m["n_sims"], noise)
        eng.check_error()
        assert digest(eng.export_tree(0)) == want and digest(eng.export_tree(1)) == want
        c = eng.counters()
        assert c["leaves_evaluated"] + c["cache_hits"] == 2 * cnt.expansions
        if rep == 1 and m["n_sims"] <= 400:
            assert c["cache_hits"] > 0.8 * 2 * cnt.expansions
    eng.close()


def test_eval_cache_whole_games_identical(azk):
    from selfplay import self_play_batch
    A, G = 49, 24
    ev = gpu_evaluator(A, "hash")
    s0, s1 = {}, {}
    plain = self_play_batch("gomoku", ev, G, 64, size=7, seed=9, stats=s0)
    cached = self_play_batch("gomoku", ev, G, 64, size=7, seed=9, stats=s1, cache_entries=1024)
    for a, b in zip(plain, cached):
        assert a.cells == b.cells and a.winner == b.winner
        assert np.stack(a.pis).tobytes() == np.stack(b.pis).tobytes() and a.qs == b.qs
    assert s0["cache_hits"] == 0 and s1["cache_hits"] > 0
    assert s1["leaves_evaluated"] + s1["cache_hits"] == s0["leaves_evaluated"]
    assert s1["cache_hits"] / s0["leaves_evaluated"] > 0.15          # the reference sees 34-66 %


@pytest.mark.parametrize("case", [3, 13, 22])
def test_shared_eval_cache_is_transparent_on_trees(azk, case):
    """cache_shared = one table for every game of the engine (the reference's MCTS.cache is process-global, mcts.py:7): entries
    are written at expansion under a claim word and read from the next tree launch on.  Transparent: every tree equals the
    oracle's; and a second search of the same positions is served (almost) entirely from the table."""
    m = next(x for x in _SMETA if x["case"] == case)
    k = f"c{m['case']}_"
    from oracle import az_oracle as ao
    game, tree, cells, player, cnt = oracle_tree(ao, m, k, ao.softmax_det)
    G = 6
    eng = azk.Engine(m["game"], G, m["n_sims"], size=m["size"] or None, cache_entries=4096, cache_shared=True)
    eng.set_positions(np.tile(cells, (G, 1)), [player] * G, [len(_SZ[k + "actions"])] * G)
    noise = torch.from_numpy(np.tile(_SZ[k + "noise"], (G, 1))).to(dev()) if m["dirichlet"] else None
    want = digest(tree.export())
    for rep in range(2):
        eng.reset_counters()
        eng.search(gpu_evaluator(game.action_dim, m["variant"]), m["n_sims"], noise)
        eng.check_error()
        for g in range(G):
            assert digest(eng.export_tree(g)) == want
        c = eng.counters()
        assert c["leaves_evaluated"] + c["cache_hits"] == G * cnt.expansions
        if rep == 1 and m["n_sims"] <= 400:
            assert c["cache_hits"] > 0.8 * G * cnt.expansions
    eng.clear_cache()
    eng.reset_counters()
    eng.search(gpu_evaluator(game.action_dim, m["variant"]), min(8, m["n_sims"]), noise)
    assert eng.counters()["cache_hits"] <= eng.counters()["leaves_evaluated"]       # cleared: the first leaves miss again
    eng.close()


def test_shared_eval_cache_whole_games_identical_and_hits_more(azk):
    """Whole self-play batches: no cache, per-game tables, one shared table - the same games, to the bit; the shared table also
    serves positions another game evaluated first, so it hits at least as often (all games open on the same board)."""
    from selfplay import self_play_batch
    A, G = 49, 48
    ev = gpu_evaluator(A, "hash")
    s0, s1, s2 = {}, {}, {}
    plain = self_play_batch("gomoku", ev, G, 64, size=7, seed=9, stats=s0)
    per_game = self_play_batch("gomoku", ev, G, 64, size=7, seed=9, stats=s1, cache_entries=1024)
    shared = self_play_batch("gomoku", ev, G, 64, size=7, seed=9, stats=s2, cache_entries=1024, cache_shared=True)
    for a, b, c in zip(plain, per_game, shared):
        assert a.cells == b.cells == c.cells and a.winner == b.winner == c.winner
        assert np.stack(a.pis).tobytes() == np.stack(b.pis).tobytes() == np.stack(c.pis).tobytes() and a.qs == b.qs == c.qs
    assert s2["leaves_evaluated"] + s2["cache_hits"] == s0["leaves_evaluated"]
    assert s2["cache_hits"] >= s1["cache_hits"] > 0
    print("hit rate per-game", s1["cache_hits"] / s0["leaves_evaluated"], "shared", s2["cache_hits"] / s0["leaves_evaluated"])


@pytest.mark.parametrize("case", [3, 10, 13, 18, 22])
@pytest.mark.parametrize("cache", ["off", "per-game", "shared"])
def test_budget_stepping_builds_the_same_trees(azk, case, cache):
    """azk_begin_search_budget: a game keeps simulating inside a launch while its simulations need no evaluator (terminal leaves,
    eval-cache hits) - fewer launches, fuller evaluator batches, and the SAME trees to the last bit (a game's simulations stay
    sequential).  Checked against the oracle's tree on golden positions (near-terminal ones included), with every cache mode."""
    m = next(x for x in _SMETA if x["case"] == case)
    k = f"c{m['case']}_"
    from oracle import az_oracle as ao
    game, tree, cells, player, cnt = oracle_tree(ao, m, k, ao.softmax_det)
    G = 5
    eng = azk.Engine(m["game"], G, m["n_sims"], size=m["size"] or None, cache_entries=0 if cache == "off" else 2048, cache_shared=cache == "shared")
    eng.set_positions(np.tile(cells, (G, 1)), [player] * G, [len(_SZ[k + "actions"])] * G)
    noise = torch.from_numpy(np.tile(_SZ[k + "noise"], (G, 1))).to(dev()) if m["dirichlet"] else None
    want = digest(tree.export())
    for rep in range(2):
        eng.reset_counters()
        launches = eng.search_budget(gpu_evaluator(game.action_dim, m["variant"]), m["n_sims"], noise, per_launch=6)
        eng.check_error()
        for g in range(G):
            assert digest(eng.export_tree(g)) == want
        c = eng.counters()
        assert c["sims"] == G * m["n_sims"]
        assert c["leaves_evaluated"] + c["cache_hits"] == G * cnt.expansions
        assert launches <= m["n_sims"] + 2
        if cnt.terminal_sims > 0.2 * m["n_sims"] or (rep == 1 and cache != "off"):
            assert launches < 0.95 * m["n_sims"], (launches, m["n_sims"])   # simulations that needed no evaluator did not cost a launch
    eng.close()


def test_budget_stepping_whole_games_identical(azk):
    from selfplay import self_play_batch
    A, G = 49, 24
    ev = gpu_evaluator(A, "hash")
    s0, s1 = {}, {}
    plain = self_play_batch("gomoku", ev, G, 64, size=7, seed=9, stats=s0, cache_entries=1024)
    fast = self_play_batch("gomoku", ev, G, 64, size=7, seed=9, stats=s1, cache_entries=1024, budget_stepping=True)
    for a, b in zip(plain, fast):
        assert a.cells == b.cells and a.winner == b.winner
        assert np.stack(a.pis).tobytes() == np.stack(b.pis).tobytes() and a.qs == b.qs
    assert s0["sims"] == s1["sims"] and s0["leaves_evaluated"] == s1["leaves_evaluated"] and s0["cache_hits"] == s1["cache_hits"]


def test_gomoku_move_lists_around_the_set_resize_thresholds(azk, ao):
    """The CPython set behind gomoku.py:93-106 grows 8 -> 32 -> 128 -> 512 slots at 5, 19 and 77 elements; the kernel replays the
    first two generations on the scalar unit and hands over to LDS rounds at the third.  Boards with exactly 1..24, 60..66 (one lane per
    key, then two chunks) and 75..80 candidate moves, several of each, against the oracle's list order."""
    size = 15
    game = ao.OracleGame("gomoku", size)
    rng = np.random.RandomState(77)
    want = set(range(3, 25)) | set(range(60, 67)) | set(range(75, 81))
    per_count, boards = {}, []
    boards.append(np.zeros((2, size, size), np.float32))          # no stone: the centre cell alone (gomoku.py:103-104)
    for k in (1, 2, 3):                                            # a full board but for k cells: 1, 2, 3 candidates
        b = np.zeros((2, size, size), np.float32)
        u = rng.rand(size, size)
        b[0] = u < 0.5
        b[1] = u >= 0.5
        for cell in rng.choice(size * size, k, replace=False):
            b[:, cell // size, cell % size] = 0.0
        assert len(game.valid_cells(b)) == k
        boards.append(b)
    tries = 0
    while tries < 40000 and any(per_count.get(m, 0) < 3 for m in want):
        tries += 1
        n_stones = int(rng.randint(1, 60))
        b = np.zeros((2, size, size), np.float32)
        if rng.rand() < 0.5:                       # clustered: few candidates per stone
            r0, c0 = rng.randint(0, size, 2)
            cells = set()
            for _ in range(4 * n_stones):               # (bounded: a window at the board's edge has fewer than n_stones cells)
                r, c = r0 + int(rng.randint(-3, 4)), c0 + int(rng.randint(-3, 4))
                if 0 <= r < size and 0 <= c < size and len(cells) < n_stones:
                    cells.add((r, c))
            cells.add((int(r0), int(c0)))
        else:                                       # scattered: many
            cells = set(map(tuple, rng.randint(0, size, (n_stones, 2)).tolist()))
        for i, (r, c) in enumerate(sorted(cells)):
            b[i & 1, r, c] = 1.0
        m = len(game.valid_cells(b))
        if m in want and per_count.get(m, 0) < 3:
            per_count[m] = per_count.get(m, 0) + 1
            boards.append(b)
    missing = sorted(m for m in want if per_count.get(m, 0) == 0)
    assert not missing, missing
    boards = np.stack(boards)
    moves, counts = azk.rules_legal_moves("gomoku", torch.from_numpy(boards).to(dev()), size)
    moves, counts = moves.cpu().numpy(), counts.cpu().numpy()
    for i in range(len(boards)):
        assert moves[i, :counts[i]].tolist() == game.valid_cells(boards[i]).tolist(), (i, int(counts[i]))


@pytest.mark.parametrize("size,plies,n_sims", [(19, 0, 40), (19, 30, 120), (20, 90, 80), (11, 14, 150)])
def test_large_and_odd_gomoku_boards_vs_oracle(azk, ao, size, plies, n_sims):
    """Boards beyond 256 cells (19x19, 20x20: seven cells per lane, the 2048-slot set tables) and an odd mid size: legal-move
    order on dense random boards and whole search trees equal the oracle's."""
    game = ao.OracleGame("gomoku", size)
    rng = np.random.RandomState(size * 1000 + plies)
    # rules: random boards of every density
    n = 24
    boards = np.zeros((n, 2, size, size), np.float32)
    for i in range(n):
        dens = rng.uniform(0.02, 0.9)
        u = rng.rand(size, size)
        boards[i, 0] = u < dens / 2
        boards[i, 1] = (u >= dens / 2) & (u < dens)
    moves, counts = azk.rules_legal_moves("gomoku", torch.from_numpy(boards).to(dev()), size)
    moves, counts = moves.cpu().numpy(), counts.cpu().numpy()
    for i in range(n):
        assert moves[i, :counts[i]].tolist() == game.valid_cells(boards[i]).tolist(), i
    # search: a random legal prefix without a winner, then a tree
    b = game.new_board()
    player, mc = 0, 0
    while mc < plies:
        vm = game.valid_cells(b)
        cell = int(vm[rng.randint(len(vm))])
        b2 = b.copy()
        nxt = game.make_move(b2, player, game.rc(cell))
        if game.check_winner(b2, player, game.rc(cell)) == -1:
            b, player, mc = b2, nxt, mc + 1
    cells = (b[0] + 2 * b[1]).astype(np.int8).reshape(-1)
    A = size * size
    noise = rng.dirichlet([0.03] * A)

    def evc(canon):
        logits, v = fixture_logits_value(torch.from_numpy(np.ascontiguousarray(canon))[None], A, "hash")
        return ao.softmax_det(logits[0].numpy()), float(v[0])
    tree = ao.OracleTree(game, cap=1 + n_sims * A)
    tree.reset(player, mc)
    ao.mcts(game, tree, b.copy(), n_sims, evc, noise, None, None, None)
    eng = azk.Engine("gomoku", 2, n_sims, size=size)
    eng.set_positions(np.tile(cells, (2, 1)), [player] * 2, [mc] * 2)
    eng.search(gpu_evaluator(A, "hash"), n_sims, torch.from_numpy(np.tile(noise, (2, 1))).to(dev()))
    eng.check_error()
    want = digest(tree.export())
    assert digest(eng.export_tree(0)) == want and digest(eng.export_tree(1)) == want
    eng.close()


@pytest.mark.parametrize("plies,variant", [(22, "uniform"), (34, "uniform"), (34, "hash")])
def test_level_scan_with_65_to_128_children_vs_oracle(azk, ao, plies, variant):
    """Late-ply 15x15 positions have 65..128 candidate moves: k_tree scans such a node two candidates per lane.  With uniform
    priors every unvisited child ties, so the tree is right only if "first maximum wins" (node.py:47) holds across the two
    candidates of a lane and across lanes; also with the hashed logits, with Dirichlet noise at the root (float64 UCB)."""
    size, n_sims = 15, 300
    game = ao.OracleGame("gomoku", size)
    rng = np.random.RandomState(4242 + plies)
    while True:
        b = game.new_board()
        player, mc = 0, 0
        # scattered stones (far apart) make many candidates out of few plies
        while mc < plies:
            cell = int(rng.randint(size * size))
            r, c = game.rc(cell)
            if b[0][r][c] or b[1][r][c]:
                continue
            b2 = b.copy()
            nxt = game.make_move(b2, player, (r, c))
            if game.check_winner(b2, player, (r, c)) == -1:
                b, player, mc = b2, nxt, mc + 1
        ncand = len(game.valid_cells(b))
        if 70 <= ncand <= 120:
            break
    cells = (b[0] + 2 * b[1]).astype(np.int8).reshape(-1)
    A = size * size
    noise = rng.dirichlet([0.03] * A)

    def evc(canon):
        logits, v = fixture_logits_value(torch.from_numpy(np.ascontiguousarray(canon))[None], A, variant)
        return ao.softmax_det(logits[0].numpy()), float(v[0])
    tree = ao.OracleTree(game, cap=1 + n_sims * A)
    tree.reset(player, mc)
    ao.mcts(game, tree, b.copy(), n_sims, evc, noise, None, None, None)
    eng = azk.Engine("gomoku", 2, n_sims, size=size)
    eng.set_positions(np.tile(cells, (2, 1)), [player] * 2, [mc] * 2)
    eng.search(gpu_evaluator(A, variant), n_sims, torch.from_numpy(np.tile(noise, (2, 1))).to(dev()))
    eng.check_error()
    want = digest(tree.export())
    assert digest(eng.export_tree(0)) == want and digest(eng.export_tree(1)) == want
    eng.close()
