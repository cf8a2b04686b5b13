"""Arena.  CPU part: compare()'s bookkeeping against a line-by-line trace of test.py:107-140's
semantics on synthetic outcomes.  GPU part: compete games equal the oracle's alternating-model games bit for bit."""
import numpy as np
import pytest
import torch

from conftest import golden_meta, load_golden
from fixture_eval import fixture_logits_value, numpy_softmax_like_reference


def reference_compare_semantics(outcomes, iterations, early_stopping):
    """Direct transcription of the reference's loop structure, kept separate from the product code."""
    win_count = [0, 0, 0]
    for i in range(iterations):
        winner = outcomes[i]
        if winner == 0:
            win_count[0 if i < iterations // 2 else 1] += 1
        elif winner == 1:
            win_count[1 if i < iterations // 2 else 0] += 1
        else:
            win_count[0] += 0.5; win_count[1] += 0.5; win_count[2] += 1
        if early_stopping:
            if win_count[1] >= int(iterations * 0.55):
                return 1
            remained_iter = iterations - (i + 1)
            if win_count[1] + remained_iter < int(iterations * 0.55):
                return 0
    return win_count[1] / iterations


def test_compare_bookkeeping():
    from arena import score_like_reference
    rng = np.random.RandomState(0)
    for _ in range(500):
        n = int(rng.choice([2, 7, 10, 50, 70]))
        outcomes = rng.choice([0, 1, -1], size=n, p=[0.45, 0.45, 0.1]).tolist()
        for es in (False, True):
            assert score_like_reference(outcomes[:n // 2], outcomes[n // 2:], n, es) == reference_compare_semantics(outcomes, n, es)


@pytest.mark.gpu
def test_compete_matches_oracle():
    from arena import compete_batch
    from oracle import az_oracle as ao
    A, G = 49, 6
    rng = np.random.RandomState(4)
    T = 49
    noise = rng.dirichlet([0.3] * A, size=(T, G))
    uniforms = rng.random_sample((T, G))
    m1 = lambda x: fixture_logits_value(x, A, "hash")
    m2 = lambda x: fixture_logits_value(x, A, "uniform")
    for sampling in (False, True):
        winners, res = compete_batch("gomoku", m1, m2, G, 40, 24, sampling=sampling, size=7,
                                     noise_fn=lambda mv: noise[mv], uniform_fn=lambda mv: uniforms[mv])
        og = ao.OracleGame("gomoku", 7)

        def ev(variant):
            def f(canon):
                logits, v = fixture_logits_value(torch.from_numpy(np.ascontiguousarray(canon))[None], A, variant)
                return ao.softmax_det(logits[0].numpy()), float(v[0])
            return f
        for g in range(G):
            out = ao.self_play(og, ev("hash"), 40, noise_fn=lambda mv: noise[mv, g], uniform_fn=lambda mv: uniforms[mv, g],
                               evaluator2=ev("uniform"), n_sims2=24, sample_until=20 if sampling else 0)
            assert res[g].cells == out["cells"].tolist() and winners[g] == out["winner"]
            assert np.stack(res[g].pis).tobytes() == out["pis"].tobytes()


@pytest.mark.gpu
def test_compare_runs_and_is_symmetric_for_identical_models():
    from arena import compare
    A = 49
    m = lambda x: fixture_logits_value(x, A, "hash")
    rate = compare("gomoku", m, m, 24, 24, 16, sampling=True, early_stopping=False, size=7, seed=2)
    assert 0.0 <= rate <= 1.0


# ---------------------------------------------------------------------------------------------------
# test.compete pinned to the reference itself (tests/golden/compete.npz, produced by running test.compete)
# ---------------------------------------------------------------------------------------------------
_CZ = load_golden("compete.npz")
_CMETA = golden_meta(_CZ)


@pytest.mark.parametrize("m", _CMETA, ids=lambda m: f"c{m['case']}-gomoku{m['size']}-{m['variant1']}-vs-{m['variant2']}")
def test_oracle_compete_equals_reference(m):
    """The oracle, driven with the reference's recorded np.random draws, ends test.compete with the reference's winner and
    final board.  The reference keeps ONE process-global MCTS.cache keyed by the position alone (mcts.py:7,38-44), so in
    a two-model game the second model is served the first model's cached evaluations: reproduced here with a shared cache."""
    from oracle import az_oracle as ao
    k = f"c{m['case']}_"
    game = ao.OracleGame("gomoku", m["size"])
    A = m["size"] ** 2
    noise, uniforms, randints = iter(_CZ[k + "noise"]), iter(_CZ[k + "uniforms"]), iter(_CZ[k + "randints"])

    def ev(variant):
        if variant is None:
            return None
        def f(canon):
            logits, v = fixture_logits_value(torch.from_numpy(np.ascontiguousarray(canon))[None], A, variant)
            return numpy_softmax_like_reference(logits[0].numpy()), float(v[0])
        return f

    def randint(n):
        want_n, val = next(randints)
        assert want_n == n
        return int(val)
    cnt = ao.Counters()
    out = ao.self_play(game, ev(m["variant1"]), m["iter1"], noise_fn=lambda mc: next(noise), uniform_fn=lambda mc: float(next(uniforms)),
                       cache=ao.OracleCache(game), randint=randint, counters=cnt, evaluator2=ev(m["variant2"]), n_sims2=m["iter2"],
                       sample_until=20 if m["sampling"] else 0)
    final = out["boards"][-1].copy()
    game.make_move(final, (len(out["boards"]) - 1) & 1, game.rc(int(out["cells"][-1])))
    assert out["winner"] == m["winner"]
    assert np.array_equal((final[0] + 2 * final[1]).astype(np.int8).reshape(-1), _CZ[k + "final_cells"])
    assert len(out["cells"]) == m["n_moves"]
    assert (cnt.mcts_count, cnt.matched) == (m["mcts_count"], m["matched"])
    for it in (noise, uniforms, randints):
        assert next(it, None) is None                                  # every recorded draw was consumed


@pytest.mark.gpu
@pytest.mark.parametrize("m", [x for x in _CMETA if x["variant1"] and x["variant2"]],
                         ids=lambda m: f"c{m['case']}-gomoku{m['size']}")
def test_gpu_compete_equals_reference(m):
    """arena.compete_batch with the reference's recorded draws and a (never evicting) shared eval cache ends with the
    reference's winner and final board - including the reference's cross-model MCTS.cache sharing."""
    from arena import compete_batch
    k = f"c{m['case']}_"
    A = m["size"] ** 2
    noise, uniforms = _CZ[k + "noise"], _CZ[k + "uniforms"]
    G = 2

    def model(variant):
        return lambda x: fixture_logits_value(x, A, variant)
    winners, res = compete_batch("gomoku", model(m["variant1"]), model(m["variant2"]), G, m["iter1"], m["iter2"], sampling=m["sampling"],
                                 size=m["size"], noise_fn=lambda mv: np.tile(noise[min(mv, len(noise) - 1)], (G, 1)),
                                 uniform_fn=lambda mv: np.full(G, uniforms[mv] if mv < len(uniforms) else 0.5), cache_entries=1 << 18)
    for g in range(G):
        assert int(winners[g]) == m["winner"]
        final = res[g].boards[-1].copy()
        r, c = res[g].actions[-1]
        final[(len(res[g].boards) - 1) & 1, r, c] = 1
        assert np.array_equal((final[0] + 2 * final[1]).astype(np.int8).reshape(-1), _CZ[k + "final_cells"]), g
