"""Pin the C oracle's board rules bit-exactly against vectors produced by the reference
(tests/golden/generate_golden.py -> rules_*.npz).  CPU only."""
import numpy as np
import pytest

from conftest import load_golden
from oracle import az_oracle as ao


def test_tuple_hash_matches_cpython():
    L = ao.lib()
    for r in range(0, 40):
        for c in range(0, 40):
            assert L.azo_py_tuple2_hash(r, c) == (hash((r, c)) & 0xFFFFFFFFFFFFFFFF)


def test_set_order_matches_cpython():
    """The restated CPython-3.10 set (probing, resizes) iterates like the interpreter's own set."""
    rng = np.random.RandomState(0)
    L = ao.lib()
    for trial in range(300):
        side = int(rng.choice([3, 7, 15, 19, 30]))
        n = int(rng.randint(1, 500))
        rs = rng.randint(0, side, n).astype(np.int32)
        cs = rng.randint(0, side, n).astype(np.int32)
        s = set()
        for r, c in zip(rs.tolist(), cs.tolist()):
            s.add((r, c))
        out = np.empty(n, np.int32)
        m = L.azo_py_set_order(rs.ctypes.data, cs.ctypes.data, n, out.ctypes.data)
        got = [(int(rs[i]), int(cs[i])) for i in out[:m]]
        assert got == list(s), (trial, side, n)


def test_tictactoe_exhaustive():
    z = load_golden("rules_ttt.npz")
    g = ao.OracleGame("tictactoe")
    assert len(z["cells"]) == 4520
    for cells, player, valid, wins, p2 in zip(z["cells"], z["player"], z["valid"], z["winner_after"], z["plane2"]):
        b = g.board_from_cells(cells, p2)
        want = [int(v) for v in valid if v >= 0]
        assert g.valid_cells(b).tolist() == want
        for cell in want:
            b2 = b.copy()
            nxt = g.make_move(b2, int(player), g.rc(cell))
            assert nxt == 1 - player
            assert b2[2, 0, 0] == 1 - player
            assert g.check_winner(b2, int(player), g.rc(cell)) == wins[cell]
            g.undo_move(b2, nxt, g.rc(cell))
            assert np.array_equal(b2[:2], b[:2]) and b2[2, 0, 0] == player


def _check_playouts(g, z):
    for gi in range(len(z["game_off"]) - 1):
        b = g.new_board()
        player = 0
        for t in range(z["game_off"][gi], z["game_off"][gi + 1]):
            want = z["valid_flat"][z["valid_off"][t]:z["valid_off"][t + 1]].tolist()
            assert g.valid_cells(b).tolist() == want, (gi, t)
            cell = int(z["actions"][t])
            nxt = g.make_move(b, player, g.rc(cell))
            assert nxt == 1 - player
            assert g.check_winner(b, player, g.rc(cell)) == z["winners"][t]
            player = nxt
        cells = (b[0] + 2 * b[1]).astype(np.int8).reshape(-1)
        assert np.array_equal(cells, z["final_cells"][gi])


def _check_random_boards(g, z, prefix="rb_"):
    cells_all = z[prefix + "cells"]
    for bi, cells in enumerate(cells_all):
        b = g.board_from_cells(cells)
        want = z[prefix + "valid_flat"][z[prefix + "valid_off"][bi]:z[prefix + "valid_off"][bi + 1]].tolist()
        assert g.valid_cells(b).tolist() == want, bi
    for bi, p, r, c, w in z[prefix + "queries"]:
        b = g.board_from_cells(cells_all[bi])
        assert g.check_winner(b, int(p), (int(r), int(c))) == w


@pytest.mark.parametrize("name,size,file", [("connect4", None, "rules_c4.npz"), ("gomoku", 7, "rules_gomoku7.npz"),
                                            ("gomoku", 15, "rules_gomoku15.npz")])
def test_playouts_and_random_boards(name, size, file):
    z = load_golden(file)
    g = ao.OracleGame(name, size)
    _check_playouts(g, z)
    _check_random_boards(g, z)


def test_tictactoe_random_boards():
    _check_random_boards(ao.OracleGame("tictactoe"), load_golden("rules_ttt_rand.npz"), prefix="")


def test_invalid_move_semantics():
    """TicTacToe/Gomoku refuse an occupied cell and return the same player (gomoku.py:57-58);
    Connect4 never checks occupancy (connect4.py:56-63)."""
    for name, size in (("tictactoe", None), ("gomoku", 7)):
        g = ao.OracleGame(name, size)
        b = g.new_board()
        assert g.make_move(b, 0, (1, 1)) == 1
        before = b.copy()
        assert g.make_move(b, 1, (1, 1)) == 1
        assert np.array_equal(b, before)
    g = ao.OracleGame("connect4")
    b = g.new_board()
    assert g.make_move(b, 0, (5, 3)) == 1
    assert g.make_move(b, 1, (5, 3)) == 0
    assert b[0, 5, 3] == 1 and b[1, 5, 3] == 1


def test_canonical_board():
    g = ao.OracleGame("gomoku", 7)
    b = g.new_board()
    b[0, 1, 2] = 1
    b[1, 3, 4] = 1
    assert np.array_equal(g.get_canonical_board(b, 0), b)
    c = g.get_canonical_board(b, 1)
    assert np.array_equal(c[0], b[1]) and np.array_equal(c[1], b[0])
    g3 = ao.OracleGame("connect4")
    b = g3.new_board()
    b[0, 5, 0] = 1
    b[2] = 1
    c = g3.get_canonical_board(b, 1)
    assert np.array_equal(c[0], b[1]) and np.array_equal(c[1], b[0]) and np.array_equal(c[2], b[2])
