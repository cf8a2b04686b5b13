"""Whole self-play games: the oracle driven with the reference's recorded RNG draws must emit the
reference's (boards, actions, pis, qs, winner) bit for bit (tests/golden/games.npz).  CPU only."""
import numpy as np
import pytest
import torch

from conftest import golden_meta, load_golden
from fixture_eval import fixture_logits_value, numpy_softmax_like_reference
from oracle import az_oracle as ao

_Z = load_golden("games.npz")
_META = golden_meta(_Z)


def _evaluator(game, variant):
    def ev(canon):
        logits, v = fixture_logits_value(torch.from_numpy(np.ascontiguousarray(canon))[None], game.action_dim, variant)
        return numpy_softmax_like_reference(logits[0].numpy()), float(v[0])
    return ev


@pytest.mark.parametrize("m", _META, ids=[f"g{m['game']}-{m['name']}{m['size']}-n{m['n_sims']}-{m['variant']}" for m in _META])
def test_self_play_bit_exact(m):
    k = f"g{m['game']}_"
    game = ao.OracleGame(m["name"], m["size"] or None)
    noise, uniforms, randints = _Z[k + "noise"], list(_Z[k + "uniforms"]), _Z[k + "randints"]
    cache = ao.OracleCache(game)
    cnt = ao.Counters()
    ri = [0]

    def randint(n):                         # replay np.random.randint draws of MCTS.simulate
        low_arg, val = randints[ri[0]]
        assert low_arg == n, (low_arg, n)
        ri[0] += 1
        return int(val)

    ui = [0]

    def uniform_fn(mc):
        u = uniforms[ui[0]]
        ui[0] += 1
        return u

    ev = _evaluator(game, m["variant"]) if m["variant"] else None
    out = ao.self_play(game, ev, m["n_sims"], noise_fn=(lambda mc: noise[mc]) if ev else None,
                       uniform_fn=uniform_fn, cache=cache if ev else None,
                       randint=randint if ev is None else None, counters=cnt)
    assert out["winner"] == m["winner"]
    assert len(out["boards"]) == m["n_moves"]
    assert out["pis"].tobytes() == _Z[k + "pis"].tobytes()
    got_cells = np.stack([(b[0] + 2 * b[1]).astype(np.int8).reshape(-1) for b in out["boards"]])
    assert np.array_equal(got_cells, _Z[k + "board_cells"])
    ref_actions = _Z[k + "actions"]
    assert out["cells"][:len(ref_actions)].tolist() == ref_actions.tolist()
    if m["name"] == "gomoku":
        assert out["qs"].tobytes() == _Z[k + "qs"].tobytes()
    assert ui[0] == len(uniforms) and ri[0] == len(randints)
    assert (cnt.mcts_count, cnt.matched, cnt.evals) == (m["mcts_count"], m["matched"], m["evals"])


@pytest.mark.parametrize("m", [x for x in _META if not x["variant"]], ids=lambda m: f"g{m['game']}-{m['name']}")
def test_vanilla_games_consume_only_the_seeded_randint_stream(m):
    """The vanilla games are a pure function of np.random.seed(seed): RandomState(seed).randint replays the recorded
    draws one for one, so the seed (not the draw tape) is enough to pin the GPU's own MT19937 + masked-rejection path."""
    k = f"g{m['game']}_"
    rs = np.random.RandomState(m["seed"])
    for n, val in _Z[k + "randints"]:
        assert rs.randint(int(n)) == val
    game = ao.OracleGame(m["name"], m["size"] or None)
    rs = np.random.RandomState(m["seed"])
    out = ao.self_play(game, None, m["n_sims"], randint=lambda n: int(rs.randint(n)))
    assert out["winner"] == m["winner"] and out["pis"].tobytes() == _Z[k + "pis"].tobytes()
