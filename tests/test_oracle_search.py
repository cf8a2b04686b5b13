"""Pin the C oracle's search (select / expand / backup / terminal handling / noise mixing) against
whole trees produced by the reference's MCTS.mcts (tests/golden/search.npz): root statistics and a
sha256 over EVERY node of the tree, bit for bit.  CPU only."""
import hashlib
import struct

import numpy as np
import pytest
import torch

from conftest import golden_meta, load_golden
from fixture_eval import fixture_logits_value, numpy_softmax_like_reference
from oracle import az_oracle as ao

_Z = load_golden("search.npz")
_META = golden_meta(_Z)


def make_evaluator(game, variant, softmax):
    def ev(canon):
        logits, v = fixture_logits_value(torch.from_numpy(np.ascontiguousarray(canon))[None], game.action_dim, variant)
        return softmax(logits[0].numpy()), float(v[0])
    return ev


def replay_position(game, actions):
    b = game.new_board()
    player = 0
    for cell in actions:
        player = game.make_move(b, player, game.rc(int(cell)))
    return b, player


def digest_of(tree):
    e = tree.export()
    h = hashlib.sha256()
    for d, c, n, w, p in zip(e["depth"], e["cell"], e["visit"], e["value"], e["prior"]):
        h.update(struct.pack("<iiqdd", int(d), int(c), int(n), float(w), float(p)))
    return h.hexdigest(), len(e["depth"])


@pytest.mark.parametrize("m", _META, ids=[f"{m['case']}-{m['game']}{m['size']}-p{m['plies']}-n{m['n_sims']}-{m['variant']}" for m in _META])
def test_search_tree_bit_exact(m):
    k = f"c{m['case']}_"
    game = ao.OracleGame(m["game"], m["size"] or None)
    board, player = replay_position(game, _Z[k + "actions"])
    assert player == m["player"]
    tree = ao.OracleTree(game)
    tree.reset(player, len(_Z[k + "actions"]))
    cache = ao.OracleCache(game)
    cnt = ao.Counters()
    noise = _Z[k + "noise"] if m["dirichlet"] else None
    before = board.copy()
    ao.mcts(game, tree, board, m["n_sims"], make_evaluator(game, m["variant"], numpy_softmax_like_reference),
            noise, cache, None, cnt)
    assert np.array_equal(board, before)                 # board restored (mcts.py contract)
    ch = tree.root_children()
    assert ch["cell"].tolist() == _Z[k + "child_cell"].tolist()
    assert ch["visit"].tolist() == _Z[k + "child_visit"].tolist()
    assert ch["value"].tobytes() == _Z[k + "child_value"].tobytes()
    assert ch["prior"].tobytes() == _Z[k + "child_prior"].tobytes()
    assert tree.root_visit == m["root_visit"] and tree.root_value == m["root_value"]
    assert tree.pi().tobytes() == _Z[k + "pi"].tobytes()
    assert digest_of(tree) == (m["digest"], m["n_nodes"])
    assert (cnt.mcts_count, cnt.matched, cnt.evals) == (m["mcts_count"], m["matched"], m["evals"])
    # bookkeeping invariants the engine's device counters are later checked against
    assert cnt.expansions + cnt.terminal_sims == m["n_sims"]
    assert cnt.edges_created == m["n_nodes"] - 1


def test_cache_is_transparent():
    """MCTS.cache changes how often the evaluator runs, never the tree."""
    m = next(x for x in _META if x["game"] == "connect4" and x["matched"] > 50)
    k = f"c{m['case']}_"
    game = ao.OracleGame("connect4")
    board, player = replay_position(game, _Z[k + "actions"])
    tree = ao.OracleTree(game)
    tree.reset(player, len(_Z[k + "actions"]))
    cnt = ao.Counters()
    ao.mcts(game, tree, board, m["n_sims"], make_evaluator(game, m["variant"], numpy_softmax_like_reference),
            _Z[k + "noise"], None, None, cnt)
    assert digest_of(tree) == (m["digest"], m["n_nodes"])
    assert cnt.matched == 0 and cnt.evals == m["evals"] + m["matched"]


def test_det_softmax_tree_matches_numpy_softmax_tree():
    """The deterministic softmax the engine shares with the oracle differs from numpy's by ulps only;
    on these cases it yields the same visit counts as the reference."""
    for m in _META:
        if m["n_sims"] > 400:
            continue
        k = f"c{m['case']}_"
        game = ao.OracleGame(m["game"], m["size"] or None)
        board, player = replay_position(game, _Z[k + "actions"])
        tree = ao.OracleTree(game)
        tree.reset(player, len(_Z[k + "actions"]))
        ao.mcts(game, tree, board, m["n_sims"], make_evaluator(game, m["variant"], ao.softmax_det),
                _Z[k + "noise"] if m["dirichlet"] else None)
        ch = tree.root_children()
        assert ch["visit"].tolist() == _Z[k + "child_visit"].tolist(), m
        np.testing.assert_allclose(ch["prior"], _Z[k + "child_prior"], rtol=1e-6, atol=0)


def _vl_tree(m, K, cache=False, counters=None):
    k = f"c{m['case']}_"
    game = ao.OracleGame(m["game"], m["size"] or None)
    board, player = replay_position(game, _Z[k + "actions"])
    tree = ao.OracleTree(game)
    tree.reset(player, len(_Z[k + "actions"]))
    before = board.copy()
    launches = ao.mcts_vl(game, tree, board, m["n_sims"], K, make_evaluator(game, m["variant"], ao.softmax_det),
                          _Z[k + "noise"] if m["dirichlet"] else None, ao.OracleCache(game) if cache else None, counters)
    assert np.array_equal(board, before)                 # the caller's board comes back as it was
    return tree, launches


@pytest.mark.parametrize("K", [2, 4])
def test_virtual_loss_oracle_invariants(K):
    """azo_mcts_vl - the sequential statement of the engine's OPT-IN K-slot virtual-loss schedule (not reference behaviour, so
    nothing of the reference pins it; the HIP kernel is compared with it bit for bit under -m gpu).  Checked here on every golden
    position: exactly n_sims simulations (root visits), visit conservation (an expanded node was visited once to expand it plus
    once per visit of a child), no virtual loss left behind (|W| <= N), determinism, the eval cache transparent, fewer launches
    than simulations, and a different tree than the sequential search (it is a different algorithm)."""
    differs = 0
    for m in _META:
        if m["n_sims"] > 400:
            continue
        cnt = ao.Counters()
        tree, launches = _vl_tree(m, K, counters=cnt)
        e = tree.export()
        depth, visit, value = e["depth"], e["visit"], e["value"]
        assert visit[0] == m["n_sims"] == cnt.mcts_count and cnt.expansions + cnt.terminal_sims == m["n_sims"]
        child_sum, has_child, stack = np.zeros(len(depth), np.int64), np.zeros(len(depth), bool), []
        for i, dpt in enumerate(depth):
            while stack and depth[stack[-1]] >= dpt:
                stack.pop()
            if stack:
                child_sum[stack[-1]] += visit[i]
                has_child[stack[-1]] = True
            stack.append(i)
        assert np.array_equal(visit[has_child], 1 + child_sum[has_child])
        assert (np.abs(value) <= visit + 1e-9).all()
        assert m["n_sims"] / K <= launches <= m["n_sims"] + 2
        assert digest_of(_vl_tree(m, K)[0]) == digest_of(tree)                       # deterministic
        assert digest_of(_vl_tree(m, K, cache=True)[0]) == digest_of(tree)           # MCTS.cache stays transparent
        differs += digest_of(tree)[0] != m["digest"]
    assert differs > 0
