"""BASELINE.json configs at FULL size on the GPU, checked through size-independent properties of the search
(visit conservation, probability normalisation, counter identities, legality of every chosen move, determinism)
plus bit-exact oracle comparisons of randomly picked games out of the batch.

  configs[1]  Connect4 6x7, 200 sims/move, 256 parallel self-play games
  configs[2]  Gomoku 15x15, 800 sims/move, 2048 parallel self-play games
"""
import numpy as np
import pytest
import torch

from fixture_eval import fixture_logits_value

pytestmark = pytest.mark.gpu


def ev(A, variant="hash"):
    return lambda x: fixture_logits_value(x, A, variant)


def oracle_search(ao, name, size, cells, player, mc, n_sims, noise_row, variant="hash"):
    game = ao.OracleGame(name, size)
    b = game.board_from_cells(cells, player)
    tree = ao.OracleTree(game)
    tree.reset(player, mc)

    def e(canon):
        logits, v = fixture_logits_value(torch.from_numpy(np.ascontiguousarray(canon))[None], game.action_dim, variant)
        return ao.softmax_det(logits[0].numpy()), float(v[0])
    ao.mcts(game, tree, b, n_sims, e, noise_row)
    return tree


def check_search_properties(eng, n_sims, noise):
    pi, q, rv = eng.root_stats()
    pi, q, rv = pi.cpu().numpy(), q.cpu().numpy(), rv.cpu().numpy()
    assert (rv == n_sims).all()                                        # root.visit == n (mcts.py:16 runs n times)
    np.testing.assert_allclose(pi.sum(1), 1.0, rtol=0, atol=1e-12)     # utils.py:54
    assert (np.round(pi * (n_sims - 1)) == pi * (n_sims - 1)).all() or np.allclose(pi * (n_sims - 1), np.round(pi * (n_sims - 1)), atol=1e-9)
    assert (np.abs(q) <= 1.0 + 1e-12).all()
    return pi, q


def test_config2_connect4_256_games_200_sims():
    import azk
    from oracle import az_oracle as ao
    from selfplay import self_play_batch
    G, n_sims, A = 256, 200, 7
    rng = np.random.RandomState(2)
    T = 42
    noise = rng.dirichlet([0.3] * A, size=(T, G))
    uniforms = rng.random_sample((T, G))
    stats = {}
    res = self_play_batch("connect4", ev(A), G, n_sims, noise_fn=lambda mv: noise[mv], uniform_fn=lambda mv: uniforms[mv], stats=stats)
    og = ao.OracleGame("connect4")
    total_plies = 0
    for g, r in enumerate(res):
        b = og.new_board()
        player = 0
        assert r.winner in (0, 1, -1)
        for t, cell in enumerate(r.cells):                              # every chosen move was legal; terminal detection exact
            assert np.array_equal(r.boards[t][:2], b[:2])
            assert cell in og.valid_cells(b).tolist()
            assert abs(r.pis[t].sum() - 1) < 1e-12 and r.pis[t][cell % 7] > 0
            mover = player
            player = og.make_move(b, player, og.rc(cell))
            w = og.check_winner(b, mover, og.rc(cell))
            last = t == len(r.cells) - 1
            assert (w != -1 or t + 1 == 42) == last
            if last:
                assert r.winner == (w if w != -1 else -1)
        total_plies += len(r.cells)
    assert stats["sims"] == total_plies * n_sims
    assert stats["leaves_evaluated"] + stats["terminal_sims"] == stats["sims"]
    # bit-exact against the oracle for a sample of the batch
    def e(canon):
        logits, v = fixture_logits_value(torch.from_numpy(np.ascontiguousarray(canon))[None], A, "hash")
        return ao.softmax_det(logits[0].numpy()), float(v[0])
    for g in (0, 17, 101, 255):
        out = ao.self_play(og, e, n_sims, noise_fn=lambda mv: noise[mv, g], uniform_fn=lambda mv: uniforms[mv, g])
        assert res[g].cells == out["cells"].tolist() and res[g].winner == out["winner"]
        assert np.stack(res[g].pis).tobytes() == out["pis"].tobytes()


def test_config3_gomoku15_2048_games_800_sims():
    import azk
    from oracle import az_oracle as ao
    G, n_sims, A = 2048, 800, 225
    eng = azk.Engine("gomoku", G, n_sims, size=15)
    eng.reset_games()
    # move 0 (empty boards, one legal move) then two more moves so the games diverge
    history = []
    for mv in range(3):
        noise, uni = eng.gen_noise(1234, 0, mv)
        cells_before, to_move, mc = eng.get_positions()
        eng.search(ev(A), n_sims, noise)
        eng.check_error()
        pi, q = check_search_properties(eng, n_sims, noise)
        history.append((cells_before.copy(), to_move.copy(), mc.copy(), noise.cpu().numpy(), pi))
        chosen, winner, done = eng.advance(uni, 8)
        chosen = chosen.cpu().numpy()
        if mv == 0:
            assert (chosen == 7 * 15 + 7).all()                        # empty board: the centre is the only child (gomoku.py:103)
            assert (pi[:, 112] == 1.0).all()
        # the chosen cell was empty and has positive visit share
        assert (cells_before[np.arange(G), chosen] == 0).all() and (pi[np.arange(G), chosen] > 0).all()
        assert (done.cpu().numpy() == 0).all()
    c = eng.counters()
    assert c["sims"] == 3 * G * n_sims and c["moves_played"] == 3 * G
    assert c["leaves_evaluated"] + c["terminal_sims"] == c["sims"]
    assert c["trace_nodes"] >= c["sims"] and c["edges_created"] > 0
    # spot checks against the oracle (whole root: order, visits, W bit-exact) on the last move's inputs
    cells_before, to_move, mc, noise_h, pi = history[-1]
    rng = np.random.RandomState(0)
    # the trees of the LAST search are gone (advance does not keep them) - redo that search on a small engine and on the oracle
    picks = rng.choice(G, 3, replace=False)
    small = azk.Engine("gomoku", len(picks), n_sims, size=15)
    small.set_positions(cells_before[picks], to_move[picks], mc[picks])
    small.search(ev(A), n_sims, torch.from_numpy(noise_h[picks]).cuda())
    pi_small = small.root_stats()[0].cpu().numpy()
    assert pi_small.tobytes() == pi[picks].tobytes()                    # same game, same inputs, any batch: same result
    for j, g in enumerate(picks):
        tree = oracle_search(ao, "gomoku", 15, cells_before[g], int(to_move[g]), int(mc[g]), n_sims, noise_h[g])
        ch, want = small.root_children(j), tree.root_children()
        assert ch["cell"].tolist() == want["cell"].tolist()
        assert ch["visit"].tolist() == want["visit"].tolist()
        assert ch["value"].tobytes() == want["value"].tobytes()
        assert tree.pi().tobytes() == pi[g].tobytes()


def test_same_seed_same_games_regardless_of_batch_split():
    """Sharding invariance on one GPU: games [0,64) as one engine == two engines of 32 with offset global indices."""
    from selfplay import self_play_batch
    A = 49
    whole = self_play_batch("gomoku", ev(A), 64, 48, size=7, seed=5, first_global_game=0)
    lo = self_play_batch("gomoku", ev(A), 32, 48, size=7, seed=5, first_global_game=0)
    hi = self_play_batch("gomoku", ev(A), 32, 48, size=7, seed=5, first_global_game=32)
    for g in range(64):
        part = lo[g] if g < 32 else hi[g - 32]
        assert whole[g].cells == part.cells and whole[g].winner == part.winner
        assert np.stack(whole[g].pis).tobytes() == np.stack(part.pis).tobytes()


def test_config2_connect4_with_the_rectangular_vit():
    """configs[1] with a real network: 256 Connect4 games x 200 sims through the bf16 ViT (6x7 tokens + cls; the
    reference's Net is square-only, so this net is build-defined - 'parity unpinned' - and only sanity is checked)."""
    from pvnet import NetConfig, PolicyValueNet
    from selfplay import self_play_batch
    cfg = NetConfig(6, 7, 3, 7, patch_size=5, embed_dim=256, num_heads=8, depth=1)
    net = PolicyValueNet(cfg, seed=0, device="cuda", dtype=torch.bfloat16, path="clsfold")
    assert net._hip is not None and net._fold is not None
    ref = PolicyValueNet(cfg, weights=net.state_dict(), device="cuda", dtype=torch.float32, path="full")
    x = (torch.rand(32, 3, 6, 7, device="cuda") < 0.2).float()
    x[:, 1] *= 1 - x[:, 0]
    torch.testing.assert_close(net(x.to(torch.bfloat16))[0], ref(x)[0], rtol=0, atol=5e-2)
    stats = {}
    res = self_play_batch("connect4", net, 256, 200, seed=4, leaf_dtype="bfloat16", stats=stats, cache_entries=1024)
    assert all(r.winner in (0, 1, -1) and 7 <= len(r.cells) <= 42 for r in res)
    assert all(abs(p.sum() - 1) < 1e-12 for r in res for p in r.pis)
    assert stats["sims"] == 200 * sum(len(r.cells) for r in res) and stats["cache_hits"] > 0


@pytest.mark.parametrize("use_graph,cache_entries", [(True, 256), (True, 0), (False, 256)])
def test_continuous_runner_equals_batched_driver_and_oracle(use_graph, cache_entries):
    """The bench's product path - SelfPlayRunner: hipGraph-replayed simulation steps with no host sync, evaluator over the
    fixed-size leaf buffer, eval cache, finished slots restarted in place - must play, slot by slot, exactly the games the
    eager batched driver plays with the same seed (and therefore the oracle's: one game is replayed on the oracle)."""
    from oracle import az_oracle as ao
    from selfplay import SelfPlayRunner, self_play_batch
    G, n_sims, size, seed = 48, 40, 7, 3
    A = size * size
    want = self_play_batch("gomoku", ev(A), G, n_sims, size=size, seed=seed)
    first = [dict(cells=[], pis=[], winner=None) for _ in range(G)]

    def on_records(move_idx, base, h_pi, h_q, h_chosen, h_winner, h_done):
        for g in range(G):
            r = first[base + g]
            if r["winner"] is not None:
                continue                                             # that slot's first game is over (the slot has restarted)
            r["cells"].append(int(h_chosen[g]))
            r["pis"].append(h_pi[g].numpy().copy())
            if int(h_done[g]):
                r["winner"] = int(h_winner[g])

    runner = SelfPlayRunner("gomoku", ev(A), G, n_sims, size=size, seed=seed, recycle=True, use_graph=use_graph,
                            cache_entries=cache_entries, on_records=on_records)
    for _ in range(size * size + 2):
        runner.play_move()
        if all(r["winner"] is not None for r in first):
            break
    runner.check_error()
    assert runner.games_finished >= G
    for g in range(G):
        assert first[g]["winner"] == want[g].winner, g
        assert first[g]["cells"] == want[g].cells, g
        assert np.stack(first[g]["pis"]).tobytes() == np.stack(want[g].pis).tobytes(), g
    c = runner.counters()
    if cache_entries:
        assert c["cache_hits"] > 0
    # one of those games on the oracle, with the engine's own noise / uniform draws for that game
    g0 = 5
    eng = runner.eng
    game = ao.OracleGame("gomoku", size)

    def ev_cpu(canon):
        logits, v = fixture_logits_value(torch.from_numpy(np.ascontiguousarray(canon))[None], A, "hash")
        return ao.softmax_det(logits[0].numpy()), float(v[0])

    def noise_fn(mc):
        return eng.gen_noise(seed, 0, mc)[0][g0].cpu().numpy()

    def uniform_fn(mc):
        return float(eng.gen_noise(seed, 0, mc)[1][g0].item())
    out = ao.self_play(game, ev_cpu, n_sims, noise_fn=noise_fn, uniform_fn=uniform_fn)
    assert out["cells"].tolist() == want[g0].cells and out["winner"] == want[g0].winner
    assert out["pis"].tobytes() == np.stack(want[g0].pis).tobytes()
