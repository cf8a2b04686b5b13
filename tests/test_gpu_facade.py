"""The reference-shaped facade on the GPU: `Game` statics and `MCTS.mcts(model, board, root, Game, n, dirichlet)`
called exactly the way the reference's callers call them (gomoku.py:134-146, test.py:41-47), checked against
golden vectors produced by the reference itself."""
import numpy as np
import pytest
import torch

from conftest import golden_meta, load_golden
from fixture_eval import fixture_logits_value

pytestmark = pytest.mark.gpu

_SZ = load_golden("search.npz")
_SMETA = golden_meta(_SZ)


class GpuFixtureModel:
    def __init__(self, A, variant):
        self.A, self.variant = A, variant

    def __call__(self, x):
        logits, v = fixture_logits_value(x, self.A, self.variant)
        return logits, v[:, None]


def game_class(m):
    from games import Connect4, Gomoku, TicTacToe
    if m["game"] == "gomoku":
        Gomoku.rows = Gomoku.cols = m["size"]                      # the reference's own way to resize
        Gomoku.action_dim = Gomoku.state_dim = m["size"] ** 2
        return Gomoku
    return {"tictactoe": TicTacToe, "connect4": Connect4}[m["game"]]


@pytest.mark.parametrize("m", [m for m in _SMETA if m["case"] in (1, 4, 9, 11, 12, 17, 19, 21, 24)],
                         ids=lambda m: f"{m['case']}-{m['game']}{m['size']}-n{m['n_sims']}")
def test_mcts_mcts_like_a_reference_caller(m):
    from ai import MCTS, Node
    Game = game_class(m)
    k = f"c{m['case']}_"
    g = Game()
    board, player = g.board, 0
    for cell in _SZ[k + "actions"]:
        mv = (int(cell) // Game.cols, int(cell) % Game.cols)
        assert mv in Game.get_valid_moves(board)
        player = Game.make_move(board, player, mv)
    assert player == m["player"]
    before = board.copy()
    root = Node(None, None, player, len(_SZ[k + "actions"]))
    np.random.seed(1000 + m["case"])                               # the generator's seed: same Dirichlet draw
    MCTS.cache.clear()                                             # the generator starts every case from an empty MCTS.cache
    count0 = MCTS.mcts_count
    MCTS.mcts(GpuFixtureModel(Game.action_dim, m["variant"]), board, root, Game, m["n_sims"], m["dirichlet"])
    assert np.array_equal(board, before)                           # board restored (mcts.py contract)
    assert MCTS.mcts_count - count0 == m["n_sims"]
    assert root.visit == m["root_visit"] and root.value == m["root_value"]
    assert [c.prevAction[0] * Game.cols + c.prevAction[1] for c in root.children] == _SZ[k + "child_cell"].tolist()
    assert [c.visit for c in root.children] == _SZ[k + "child_visit"].tolist()
    assert np.array([c.value for c in root.children]).tobytes() == _SZ[k + "child_value"].tobytes()
    np.testing.assert_allclose([float(c.prior) for c in root.children], _SZ[k + "child_prior"], rtol=1e-6)
    assert isinstance(root.children[0].prior, np.float64 if m["dirichlet"] else np.float32)
    # what callers read afterwards
    best = root.max_visit_child()
    assert best.visit == max(_SZ[k + "child_visit"]) and best is next(c for c in root.children if c.visit == best.visit)
    pi = root.visit_distribution(Game)
    assert pi.tobytes() == _SZ[k + "pi"].tobytes()
    assert all(c.currentPlayer == 1 - player and c.move_count == root.move_count + 1 and c.parent is root for c in root.children)
    assert "Visit" in root.children[0].to_string(Game)


def test_game_statics_behave_like_the_reference():
    from games import Connect4, Gomoku, TicTacToe
    Gomoku.rows = Gomoku.cols = 7
    Gomoku.action_dim = Gomoku.state_dim = 49
    g = Gomoku()
    assert Gomoku.get_valid_moves(g.board) == [(3, 3)]              # empty board: centre only (gomoku.py:103-104)
    assert Gomoku.make_move(g.board, 0, (3, 3)) == 1
    assert Gomoku.get_valid_moves(g.board) == [(4, 4), (2, 4), (3, 4), (4, 3), (4, 2), (2, 3), (2, 2), (3, 2)]  # CPython set order
    assert Gomoku.make_move(g.board, 1, (3, 3)) == 1                # occupied: unchanged player
    assert Gomoku.check_winner(g.board, 0, (3, 3)) == -1
    c = Gomoku.get_canonical_board(g.board, 1)
    assert c[1, 3, 3] == 1 and c[0].sum() == 0 and Gomoku.get_canonical_board(g.board, 0) is g.board
    Gomoku.undo_move(g.board, 1, (3, 3))
    assert g.board.sum() == 0
    t = TicTacToe()
    for i, mv in enumerate([(0, 0), (1, 0), (0, 1), (1, 1)]):
        assert TicTacToe.make_move(t.board, i % 2, mv) == 1 - i % 2
    assert t.board[2, 0, 0] == 0                                    # side-to-move plane (tictactoe.py:41)
    assert TicTacToe.make_move(t.board, 0, (0, 2)) == 1 and TicTacToe.check_winner(t.board, 0, (0, 2)) == 0
    c4 = Connect4()
    assert Connect4.get_valid_moves(c4.board) == [(5, c) for c in range(7)]
    assert Connect4.make_move(c4.board, 0, (None, 3)) == 0          # full column: unchanged (connect4.py:57-60)


def test_gomoku_self_play_returns_the_reference_tuple():
    from games import Gomoku
    Gomoku.rows = Gomoku.cols = 7
    Gomoku.action_dim = Gomoku.state_dim = 49
    np.random.seed(3)
    boards, actions, pis, qs, winner = Gomoku().self_play(GpuFixtureModel(49, "hash"), 40)
    assert actions[0] == (-1, -1) and len(actions) == len(boards) + 1 == len(pis) + 1 == len(qs) + 1
    assert winner in (0, 1, -1) and boards[0].shape == (2, 7, 7) and boards[0].sum() == 0
    assert all(abs(p.sum() - 1) < 1e-12 for p in pis) and actions[1] == (3, 3)
    for i in range(1, len(boards)):                                 # raw boards, one more stone each ply
        assert boards[i].sum() == i


@pytest.mark.parametrize("gi", [6, 7])
def test_vanilla_self_play_like_a_reference_caller(gi):
    """np.random.seed(s); Game().self_play(None, n) - the reference's own vanilla call (config 1 of BASELINE.json) -
    returns the reference's game bit for bit AND leaves np.random where the reference leaves it."""
    from games import Connect4, TicTacToe
    z = load_golden("games.npz")
    m = next(x for x in golden_meta(z) if x["game"] == gi)
    Game = {"tictactoe": TicTacToe, "connect4": Connect4}[m["name"]]
    k = f"g{gi}_"
    np.random.seed(m["seed"])
    boards, pis, winner = Game().self_play(None, m["n_sims"])
    assert winner == m["winner"] and len(boards) == m["n_moves"]
    assert np.stack(pis).tobytes() == z[k + "pis"].tobytes()
    got_cells = np.stack([(b[0] + 2 * b[1]).astype(np.int8).reshape(-1) for b in boards])
    assert np.array_equal(got_cells, z[k + "board_cells"])
    after = np.random.randint(1 << 30)
    rs = np.random.RandomState(m["seed"])
    for n, val in z[k + "randints"]:
        assert rs.randint(int(n)) == val
    assert after == rs.randint(1 << 30)
    if gi == 6:     # first-move pi of the seed-0 TicTacToe 25-sim vanilla game observed on the reference
        np.testing.assert_allclose(pis[0], [.1667, .125, .0417, .0417, .375, .0417, .0417, .0833, .0833], atol=5e-5)


def test_vanilla_mcts_mcts_reference_signature():
    """MCTS.mcts(None, board, root, Game, n): root mutated like the reference's (children order, visit, value, prior 0)."""
    from ai import MCTS, Node
    from games import Gomoku
    from oracle import az_oracle as ao
    Gomoku.rows = Gomoku.cols = 7
    Gomoku.action_dim = Gomoku.state_dim = 49
    g = Gomoku()
    player = 0
    for mv in [(3, 3), (2, 4), (4, 4), (3, 5)]:
        player = Gomoku.make_move(g.board, player, mv)
    before = g.board.copy()
    og = ao.OracleGame("gomoku", 7)
    tree = ao.OracleTree(og, cap=1 + 90 * 49)
    tree.reset(player, 4)
    rs = np.random.RandomState(21)
    ob = before.copy()
    ao.mcts(og, tree, ob, 90, None, None, None, lambda n: int(rs.randint(n)))
    np.random.seed(21)
    root = Node(None, None, player, 4)
    MCTS.mcts(None, g.board, root, Gomoku, 90)
    assert np.array_equal(g.board, before)
    want = tree.root_children()
    assert [og.cell(c.prevAction) for c in root.children] == want["cell"].tolist()
    assert [c.visit for c in root.children] == want["visit"].tolist()
    assert [c.value for c in root.children] == want["value"].tolist()
    assert all(c.prior == 0.0 for c in root.children)
    assert root.visit == 90 and root.value == tree.root_value
    assert np.random.randint(1 << 30) == rs.randint(1 << 30)


def test_train_py_facade_collect_and_train():
    """train.collect_data / save_data_to_buffer / train with the reference's signatures (train.py:30-123): the device ring
    filled by the engine holds exactly what the host path (save_data_to_buffer on the same games) appends, and one train
    call updates the network in place."""
    import azk
    import train as az_train
    from games import Gomoku
    from pvnet import NetConfig, PolicyValueNet
    Gomoku.rows = Gomoku.cols = 7
    Gomoku.action_dim = Gomoku.state_dim = 49
    model = GpuFixtureModel(49, "hash")
    dev_buf = azk.DeviceReplay(4096, 2, 7, 7, 49)

    class HostBuffer:                                              # the reference's ReplayBuffer interface (replay_buffer.py:7-13)
        def __init__(self):
            self.buffer = []

        def add(self, s, p, r):
            self.buffer.append((np.array(s, np.float32), np.array(p, np.float64), list(r)))

        def size(self):
            return len(self.buffer)
    host_buf = HostBuffer()
    r1 = az_train.collect_data(Gomoku, model, dev_buf, 6, 30, seed=11)
    r2 = az_train.collect_data(Gomoku, model, host_buf, 6, 30, seed=11)
    assert r1 == r2 and sum(r1) == 6
    assert dev_buf.size() == host_buf.size() > 0
    # same multiset of tuples (the device ring orders games by finishing time, the host path by game index)
    def keyed(items):
        return sorted((s.tobytes(), p.tobytes(), float(z[0])) for s, p, z in items)
    assert keyed(dev_buf.to_reference_deque()) == keyed(host_buf.buffer)
    # train(): in-place weight update of a PolicyValueNet from the device ring
    cfg = NetConfig(7, 7, 2, 49, 3, 128, 4, 1)
    net = PolicyValueNet(cfg, seed=0, device="cuda", dtype=torch.bfloat16, path="cls")
    before = net.state_dict()
    out = az_train.train(net, 64, dev_buf, 3, 0.00025, "cuda")
    assert len(out) == 4 and all(np.isfinite(v) for v in out)
    after = net.state_dict()
    assert any(not torch.equal(before[k], after[k]) for k in before)
    logits, v = net(torch.zeros(2, 2, 7, 7, device="cuda"))
    assert logits.shape == (2, 49) and torch.isfinite(logits.float()).all()


@pytest.mark.parametrize("gi", [0, 1, 4, 5])
def test_network_self_play_like_a_seeded_reference_caller(gi):
    """np.random.seed(s); Game().self_play(model, n) - the reference's own call (train.collect_data) - returns the
    reference's game: the facade draws np.random.dirichlet / np.random.choice from the global stream exactly where the
    reference does, so the recorded golden games come out bit for bit (pis, boards, actions, winner)."""
    from games import Connect4, Gomoku, TicTacToe
    z = load_golden("games.npz")
    m = next(x for x in golden_meta(z) if x["game"] == gi)
    if m["name"] == "gomoku":
        Gomoku.rows = Gomoku.cols = m["size"]
        Gomoku.action_dim = Gomoku.state_dim = m["size"] ** 2
    Game = {"gomoku": Gomoku, "tictactoe": TicTacToe, "connect4": Connect4}[m["name"]]
    k = f"g{gi}_"
    from ai import MCTS
    MCTS.cache.clear()            # a different model from the previous test: the reference clears its global cache too (main.py:55)
    np.random.seed(m["seed"])
    out = Game().self_play(GpuFixtureModel(Game.action_dim, m["variant"]), m["n_sims"])
    if m["name"] == "gomoku":
        boards, actions, pis, qs, winner = out
        assert [a[0] * Game.cols + a[1] for a in actions[1:]] == z[k + "actions"].tolist()
        assert np.array(qs, np.float64).tobytes() == z[k + "qs"].tobytes()
    else:
        boards, pis, winner = out
    assert winner == m["winner"] and len(boards) == m["n_moves"]
    assert np.stack(pis).tobytes() == z[k + "pis"].tobytes()
    got_cells = np.stack([(b[0] + 2 * b[1]).astype(np.int8).reshape(-1) for b in boards])
    assert np.array_equal(got_cells, z[k + "board_cells"])
    # and the global stream is where the reference left it: the next draw equals the one after replaying the recorded draws
    after = np.random.random_sample()
    rs = np.random.RandomState(m["seed"])
    A = Game.action_dim
    ui = 0
    for t in range(len(z[k + "noise"])):
        rs.dirichlet([0.03] * A)
        if ui < len(z[k + "uniforms"]) and (m["name"] != "gomoku" or t < 8):
            rs.random_sample()
            ui += 1
    assert after == rs.random_sample()


@pytest.mark.parametrize("ci", [0, 1, 2, 3])
def test_compete_like_a_seeded_reference_caller(ci):
    """arena.compete(Game, model1, model2, ...) = test.compete (test.py:60-105) on the global np.random stream and one shared
    eval cache: the reference's recorded games (tests/golden/compete.npz) come out exactly - winner, final board, the
    MCTS.mcts_count / MCTS.matched bookkeeping - including a vanilla (None) side."""
    from ai import MCTS
    from arena import compete
    from games import Gomoku
    z = load_golden("compete.npz")
    m = golden_meta(z)[ci]
    Gomoku.rows = Gomoku.cols = m["size"]
    Gomoku.action_dim = Gomoku.state_dim = m["size"] ** 2
    A = m["size"] ** 2
    m1 = GpuFixtureModel(A, m["variant1"]) if m["variant1"] else None
    m2 = GpuFixtureModel(A, m["variant2"]) if m["variant2"] else None
    MCTS.cache.clear()
    MCTS.matched = 0
    MCTS.mcts_count = 0
    np.random.seed(m["seed"])
    winner, board = compete(Gomoku, m1, m2, m["iter1"], m["iter2"], sampling=m["sampling"])
    assert winner == m["winner"]
    assert np.array_equal((board[0] + 2 * board[1]).astype(np.int8).reshape(-1), z[f"c{ci}_final_cells"])
    assert (MCTS.mcts_count, MCTS.matched) == (m["mcts_count"], m["matched"])


@pytest.mark.parametrize("ki", [0, 1, 2])
def test_compare_like_a_seeded_reference_caller(ki):
    """arena.compare_sequential = test.compare (test.py:107-140): same games in the same order on the global np.random
    stream with MCTS.cache kept across them - the reference's per-game winners, returned value (incl. an early stop) and
    MCTS.mcts_count / MCTS.matched totals (tests/golden/compete.npz, produced by running test.compare)."""
    import json
    from ai import MCTS
    from arena import compare_sequential
    from games import Gomoku
    z = load_golden("compete.npz")
    m = json.loads(bytes(z["compare_meta_json"]).decode())[ki]
    Gomoku.rows = Gomoku.cols = m["size"]
    Gomoku.action_dim = Gomoku.state_dim = m["size"] ** 2
    A = m["size"] ** 2
    best = GpuFixtureModel(A, m["best"]) if m["best"] else None
    cont = GpuFixtureModel(A, m["contender"]) if m["contender"] else None
    MCTS.cache.clear()
    MCTS.matched = 0
    MCTS.mcts_count = 0
    np.random.seed(m["seed"])
    winners = []
    value = compare_sequential(Gomoku, best, cont, m["best_iter"], m["contender_iter"], m["iterations"], m["sampling"],
                               m["early_stopping"], winners_out=winners)
    assert winners == m["winners"]
    assert float(value) == m["value"]
    assert (MCTS.mcts_count, MCTS.matched) == (m["mcts_count"], m["matched"])


def test_collect_data_like_a_seeded_reference_caller():
    """train.collect_data(..., batched=False): Game().self_play + save_data_to_buffer game by game on the global stream - the
    buffer of a seeded call is the reference's (sha256 recorded from the reference's ReplayBuffer, games.npz)."""
    import hashlib, struct
    import train as az_train
    from ai import MCTS
    from games import Gomoku
    z = load_golden("games.npz")
    m = next(x for x in golden_meta(z) if x["game"] == 0)
    Gomoku.rows = Gomoku.cols = m["size"]
    Gomoku.action_dim = Gomoku.state_dim = m["size"] ** 2

    class HostBuffer:
        def __init__(self):
            self.buffer = []

        def add(self, s, p, r):
            self.buffer.append((np.array(s, np.float32), np.array(p, np.float64), list(r)))

        def size(self):
            return len(self.buffer)
    buf = HostBuffer()
    MCTS.cache.clear()
    np.random.seed(m["seed"])
    res = az_train.collect_data(Gomoku, GpuFixtureModel(49, m["variant"]), buf, 1, m["n_sims"], batched=False)
    assert sum(res) == 1 and res[m["winner"] if m["winner"] >= 0 else 2] == 1
    h = hashlib.sha256()
    for s, p, zz in buf.buffer:
        h.update(np.ascontiguousarray(s, np.float32).tobytes())
        h.update(np.ascontiguousarray(p, np.float64).tobytes())
        h.update(struct.pack("<d", float(zz[0])))
    assert buf.size() == m["buffer_len"] and h.hexdigest() == m["buffer_digest"]
