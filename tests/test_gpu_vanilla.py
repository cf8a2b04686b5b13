"""GPU parity tests of vanilla mode (model=None: UCB1 walk, expansion without priors, random rollouts - mcts.py:57-79,
utils.py:29-44 'normal') through the C ABI.  The engine's per-game MT19937 + numpy masked-rejection randint must consume
exactly the reference's np.random stream, so a game seeded with RandomState(seed) reproduces the reference's recorded
vanilla games (tests/golden/games.npz) and the C oracle driven by numpy's own randint, bit for bit."""
import hashlib
import struct

import numpy as np
import pytest
import torch

from conftest import golden_meta, load_golden

pytestmark = pytest.mark.gpu

_Z = load_golden("games.npz")
_META = golden_meta(_Z)


@pytest.fixture(scope="module")
def azk():
    import azk as m
    m.lib()
    return m


@pytest.fixture(scope="module")
def ao():
    from oracle import az_oracle
    return az_oracle


def digest(e):
    h = hashlib.sha256()
    for d, c, n, w, p in zip(e["depth"], e["cell"], e["visit"], e["value"], e["prior"]):
        h.update(struct.pack("<iiqdd", int(d), int(c), int(n), float(w), float(p)))
    return h.hexdigest(), len(e["depth"])


def test_device_mt19937_randint_stream_matches_numpy(azk):
    """Searches advance the device generator exactly as np.random.randint advances numpy's (checked through the state)."""
    m = next(x for x in _META if x["name"] == "tictactoe" and not x["variant"])
    eng = azk.Engine("tictactoe", 2, 64)
    rs = np.random.RandomState(1234)
    for _ in range(700):                      # move the position close to a regeneration boundary and beyond
        rs.randint(7)
    st0 = azk.mt_state_from_numpy(rs.get_state())
    eng.vanilla_set_rng(np.stack([st0, st0]))
    assert np.array_equal(eng.vanilla_get_rng(), np.stack([st0, st0]))
    eng.reset_games()
    eng.vanilla_search(64, chunk=10)
    eng.check_error()
    st1 = eng.vanilla_get_rng()
    assert np.array_equal(st1[0], st1[1]) and not np.array_equal(st1[0], st0)
    eng.close()


@pytest.mark.parametrize("m", [x for x in _META if not x["variant"]], ids=lambda m: f"g{m['game']}-{m['name']}-n{m['n_sims']}")
def test_vanilla_golden_games(azk, m):
    """Reference `Game().self_play(None, n)` under np.random.seed(seed): pis, boards, moves, winner and the final
    generator state (numpy's after replaying the recorded draws) - all bit-exact."""
    k = f"g{m['game']}_"
    G = 3
    eng = azk.Engine(m["name"], G, m["n_sims"], size=m["size"] or None)
    rs = np.random.RandomState(m["seed"])
    eng.vanilla_set_rng(np.tile(azk.mt_state_from_numpy(rs.get_state()), (G, 1)))
    eng.reset_games()
    pis, boards, cells_played = [], [], []
    winner = None
    for t in range(m["n_moves"] + 2):
        eng.vanilla_search(m["n_sims"], chunk=7)
        pi, q, rv = eng.root_stats()
        assert rv.tolist() == [m["n_sims"]] * G
        pis.append(pi[1].cpu().numpy().copy())
        boards.append(eng.get_positions()[0][2].copy())
        chosen, win, done = eng.advance(None)                    # model=None => max_visit_child (tictactoe.py:117)
        cells_played.append(int(chosen[0].item()))
        if int(done[0].item()):
            winner = int(win[0].item())
            break
    eng.check_error()
    assert winner == m["winner"] and len(pis) == m["n_moves"]
    assert np.stack(pis).tobytes() == _Z[k + "pis"].tobytes()
    assert np.array_equal(np.stack(boards), _Z[k + "board_cells"])
    ref_actions = _Z[k + "actions"]
    assert cells_played[:len(ref_actions)] == ref_actions.tolist()
    for n, val in _Z[k + "randints"]:
        assert rs.randint(int(n)) == val
    want = azk.mt_state_from_numpy(rs.get_state())
    got = eng.vanilla_get_rng()
    for g in range(G):
        assert np.array_equal(got[g], want), g
    c = eng.counters()
    assert c["sims"] == G * m["mcts_count"]
    eng.close()


@pytest.mark.parametrize("name,size,n_sims,moves,seed", [("gomoku", 7, 120, 6, 3), ("gomoku", 15, 60, 9, 4), ("gomoku", 7, 300, 0, 5),
                                                        ("connect4", None, 150, 11, 6), ("tictactoe", None, 200, 3, 7)])
def test_vanilla_tree_equals_oracle(azk, ao, name, size, n_sims, moves, seed):
    """Whole tree after a vanilla search from a mid-game position == the oracle's (driven by numpy's randint)."""
    game = ao.OracleGame(name, size)
    rng = np.random.RandomState(100 + seed)
    b = game.new_board()
    player, mc = 0, 0
    for _ in range(moves):                                       # random legal prefix without a winner
        for _try in range(50):
            vm = game.valid_cells(b)
            cell = int(vm[rng.randint(len(vm))])
            b2 = b.copy()
            nxt = game.make_move(b2, player, game.rc(cell))
            if game.check_winner(b2, player, game.rc(cell)) == -1:
                b, player, mc = b2, nxt, mc + 1
                break
    cells = (b[0] + 2 * b[1]).astype(np.int8).reshape(-1)
    rs = np.random.RandomState(seed)
    st = azk.mt_state_from_numpy(rs.get_state())
    tree = ao.OracleTree(game, cap=1 + n_sims * game.rows * game.cols)
    tree.reset(player, mc)
    cnt = ao.Counters()
    board_before = b.copy()
    ao.mcts(game, tree, b, n_sims, None, None, None, lambda n: int(rs.randint(n)), cnt)
    assert np.array_equal(b, board_before)
    G = 2
    eng = azk.Engine(name, G, n_sims, size=size)
    eng.set_positions(np.tile(cells, (G, 1)), [player] * G, [mc] * G)
    eng.vanilla_set_rng(np.tile(st, (G, 1)))
    eng.vanilla_search(n_sims, chunk=50)
    eng.check_error()
    want = digest(tree.export())
    for g in range(G):
        assert digest(eng.export_tree(g)) == want, g
    assert np.array_equal(eng.vanilla_get_rng()[1], azk.mt_state_from_numpy(rs.get_state()))
    c = eng.counters()
    assert c["edges_scanned"] == G * cnt.edges_scanned and c["trace_nodes"] == G * cnt.trace_nodes
    assert c["edges_created"] == G * cnt.edges_created and c["terminal_sims"] == G * cnt.terminal_sims
    eng.close()


def test_compete_against_vanilla_equals_oracle(azk, ao):
    """test.compare(Game, None, model, ...) (main.py:76): vanilla MCTS plays one side inside the batched engine;
    every game equals the oracle's alternating (None, model) game driven by the same numpy stream."""
    from fixture_eval import fixture_logits_value
    G = 3
    game = ao.OracleGame("gomoku", 7)

    def ev_gpu(x):
        return fixture_logits_value(x, 49, "hash")

    def ev_cpu(canon):
        logits, v = fixture_logits_value(torch.from_numpy(np.ascontiguousarray(canon))[None], 49, "hash")
        return ao.softmax_det(logits[0].numpy()), float(v[0])
    states = np.stack([azk.mt_state_from_numpy(np.random.RandomState(50 + g).get_state()) for g in range(G)])
    from selfplay import self_play_batch
    res = self_play_batch("gomoku", (None, ev_gpu), G, (40, 30), size=7, dirichlet=False, sample_until=0, vanilla_rng=states)
    for g in range(G):
        rs = np.random.RandomState(50 + g)
        out = ao.self_play(game, None, 40, randint=lambda n: int(rs.randint(n)), evaluator2=ev_cpu,
                           n_sims2=30, sample_until=0)
        assert out["winner"] == res[g].winner
        assert out["cells"].tolist() == res[g].cells
        assert out["pis"].tobytes() == np.stack(res[g].pis).tobytes()


def test_compare_against_vanilla_with_a_game_class(azk):
    """test.compare(Game, None, model, ...) as main.py:76 calls it: a Game class and a None (vanilla) opponent."""
    from arena import compare
    from fixture_eval import fixture_logits_value
    from games import Gomoku
    Gomoku.rows = Gomoku.cols = 7
    Gomoku.action_dim = Gomoku.state_dim = 49
    rate = compare(Gomoku, None, lambda x: fixture_logits_value(x, 49, "hash"), 20, 20, iterations=4, sampling=False, early_stopping=False)
    assert 0.0 <= rate <= 1.0
