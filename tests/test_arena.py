"""Arena.  CPU part: compare()'s bookkeeping against a line-by-line trace of test.py:107-140's
semantics on synthetic outcomes.  GPU part: compete games equal the oracle's alternating-model games bit for bit."""
import numpy as np
import pytest
import torch

from fixture_eval import fixture_logits_value


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
