"""OPT-IN virtual-loss expansion (north_star: "virtual-loss expansion"): K leaves in flight per game.  A
selection leaves a visit and a lost game on its path until the leaf's value is backed up, so the same game's next selections go
elsewhere; the evaluator batch holds up to G * K boards per launch.  It is NOT the reference's search (ai/mcts.py:16-60 is strictly
sequential) - results differ from parity mode by design - so it is pinned to a SEQUENTIAL RESTATEMENT of its schedule in the
oracle (oracle/az_oracle.c azo_mcts_vl: whole trees bit for bit) and tested through its own invariants:
  K = 1 is the parity mode bit for bit; visit conservation; no virtual loss left behind; legality; determinism."""
import numpy as np
import pytest
import torch

from conftest import golden_meta, load_golden
from fixture_eval import fixture_logits_value

pytestmark = pytest.mark.gpu

_SZ = load_golden("search.npz")
_SMETA = golden_meta(_SZ)


def _ev(A, variant="hash"):
    return lambda x: fixture_logits_value(x, A, variant)


def _tree_checks(t, n_sims):
    """visit conservation on a preorder export: an expanded node was visited once to expand it plus once per visit of a child;
    |W| <= N everywhere (a virtual loss left behind would break one of the two)."""
    depth, visit, value = t["depth"], t["visit"], t["value"]
    assert visit[0] == n_sims
    child_sum = np.zeros(len(depth), np.int64)
    has_child = np.zeros(len(depth), bool)
    stack = []
    for i, dpt in enumerate(depth):
        while stack and depth[stack[-1]] >= dpt:
            stack.pop()
        if stack:
            child_sum[stack[-1]] += visit[i]
            has_child[stack[-1]] = True
        stack.append(i)
    exp = has_child
    assert np.array_equal(visit[exp], 1 + child_sum[exp])
    assert (np.abs(value) <= visit + 1e-9).all()


@pytest.mark.parametrize("K", [1, 2, 4])
@pytest.mark.parametrize("case", [3, 13, 22])
def test_invariants_and_k1_is_parity(K, case):
    import azk
    from test_gpu_engine import digest, oracle_tree
    from oracle import az_oracle as ao
    m = next(x for x in _SMETA if x["case"] == case)
    k = f"c{m['case']}_"
    game, tree, cells, player, cnt = oracle_tree(ao, m, k, ao.softmax_det)
    G, n = 4, m["n_sims"]
    noise = torch.from_numpy(np.tile(_SZ[k + "noise"], (G, 1))).cuda() if m["dirichlet"] else None
    digs = []
    for rep in range(2):
        eng = azk.Engine(m["game"], G, n, size=m["size"] or None, leaves_per_step=K, cache_entries=512 if rep else 0)
        assert eng.slots == G * K and eng.leaf_boards.shape[0] == G * K
        eng.set_positions(np.tile(cells, (G, 1)), [player] * G, [len(_SZ[k + "actions"])] * G)
        launches = eng.search_budget(_ev(game.action_dim, m["variant"]), n, noise)
        eng.check_error()
        c = eng.counters()
        assert c["sims"] == G * n                                   # exactly the budget of completed simulations
        pi, q, visits = eng.root_stats()
        assert torch.allclose(pi.sum(1), torch.ones(G, dtype=pi.dtype, device=pi.device))
        trees = [eng.export_tree(g) for g in range(G)]
        for t in trees:
            _tree_checks(t, n)
        assert len({digest(t) for t in trees}) == 1                 # same position, same noise: the slots of every game behave alike
        digs.append(digest(trees[0]))
        if K == 1:
            assert digest(trees[0]) == digest(tree.export())        # K = 1: the reference's tree, bit for bit
        else:
            assert launches < n                                     # K simulations per launch
        # legality: the root's children are exactly the position's legal moves
        kids = trees[0]["cell"][trees[0]["depth"] == 1]
        legal = ao.OracleGame(m["game"], m["size"] or None) if False else None
        assert len(set(kids.tolist())) == len(kids)
        eng.close()
    assert digs[0] == digs[1]                                       # deterministic, and the eval cache stays transparent


@pytest.mark.parametrize("K", [2, 4])
def test_trees_equal_the_sequential_statement_of_the_schedule(K):
    """the K-slot schedule of k_tree<.., MULTI> against its sequential restatement in the oracle
    (oracle/az_oracle.c azo_mcts_vl: slot k expands its pending leaf, then selects with N += 1, W -= 1 on the path; a walk that
    ends on a node whose expansion is pending gives up; exactly n_sims simulations) - whole trees, every node (depth, action, N,
    W, P) bit for bit, on every golden position (three games, with and without root noise, tie-heavy evaluators, near-terminal
    positions), with the eval cache off and on."""
    import azk
    from test_gpu_engine import digest
    from oracle import az_oracle as ao
    checked = 0
    for m in _SMETA:
        if m["n_sims"] > 400:
            continue
        k = f"c{m['case']}_"
        game = ao.OracleGame(m["game"], m["size"] or None)
        b, player = game.new_board(), 0
        for cell in _SZ[k + "actions"]:
            player = game.make_move(b, player, game.rc(int(cell)))
        tree = ao.OracleTree(game)
        tree.reset(player, len(_SZ[k + "actions"]))

        def ev(canon, game=game, m=m):
            logits, v = fixture_logits_value(torch.from_numpy(np.ascontiguousarray(canon))[None], game.action_dim, m["variant"])
            return ao.softmax_det(logits[0].numpy()), float(v[0])
        launches = ao.mcts_vl(game, tree, b, m["n_sims"], K, ev, _SZ[k + "noise"] if m["dirichlet"] else None)
        want = digest(tree.export())
        cells = (b[0] + 2 * b[1]).astype(np.int8).reshape(-1)
        G = 2
        noise = torch.from_numpy(np.tile(_SZ[k + "noise"], (G, 1))).cuda() if m["dirichlet"] else None
        for entries in (0, 256):
            eng = azk.Engine(m["game"], G, m["n_sims"], size=m["size"] or None, leaves_per_step=K, cache_entries=entries)
            eng.set_positions(np.tile(cells, (G, 1)), [player] * G, [len(_SZ[k + "actions"])] * G)
            got_launches = eng.search_budget(_ev(game.action_dim, m["variant"]), m["n_sims"], noise)
            eng.check_error()
            for g in range(G):
                assert digest(eng.export_tree(g)) == want, (m["case"], K, entries)
            if entries == 0:
                assert got_launches == launches, (m["case"], K, got_launches, launches)
            eng.close()
        checked += 1
    assert checked >= 15


def test_whole_games_and_runner():
    """Whole self-play games with two leaves in flight: they end legally (winner or draw), deterministically; and the graph runner
    plays with the real network, counting exactly n_sims completed simulations per search."""
    from pvnet import NetConfig, PolicyValueNet
    from selfplay import SelfPlayRunner, self_play_batch
    A, G = 49, 16
    a = self_play_batch("gomoku", _ev(A), G, 48, size=7, seed=3, leaves_per_step=2)
    b = self_play_batch("gomoku", _ev(A), G, 48, size=7, seed=3, leaves_per_step=2)
    ref = self_play_batch("gomoku", _ev(A), G, 48, size=7, seed=3)
    assert all(x.cells == y.cells and x.winner == y.winner for x, y in zip(a, b))
    assert all(r.winner in (-1, 0, 1) and len(set(r.cells)) == len(r.cells) for r in a)      # every move on an empty cell
    assert any(x.cells != y.cells for x, y in zip(a, ref))          # it is a different search than the reference's: opt-in
    cfg = NetConfig(15, 15, 2, 225, 5, 512, 8, 1)
    net = PolicyValueNet(cfg, seed=6, device="cuda", dtype=torch.bfloat16, path="clsfold")
    r = SelfPlayRunner("gomoku", net, 64, 96, size=15, seed=9, leaf_dtype="bfloat16", recycle=True, use_graph=True, cache_entries=256,
                       cache_shared=True, leaves_per_step=2)
    for _ in range(3):
        r.play_move()
    r.check_error()
    assert r.counters()["sims"] == 3 * 64 * 96
    assert r.launches < 0.75 * 3 * 96
