"""(state, pi, z) emission on the GPU (azk_emit_finished): whole games replayed with the reference's recorded RNG
draws must put into the device ring exactly the tuples train.save_data_to_buffer produced (sha256 recorded from the
reference in tests/golden/games.npz), in the reference's order; plus ring / sampling behaviour."""
import numpy as np
import pytest
import torch

from conftest import golden_meta, load_golden
from fixture_eval import fixture_logits_value

pytestmark = pytest.mark.gpu

_GZ = load_golden("games.npz")
_META = [m for m in golden_meta(_GZ) if "buffer_digest" in m and m["variant"]]


@pytest.mark.parametrize("m", _META, ids=[f"g{m['game']}-gomoku{m['size']}-n{m['n_sims']}-{m['variant']}" for m in _META])
def test_emitted_tuples_match_reference_buffer(m):
    import azk
    from oracle import replay_oracle as ro
    from selfplay import self_play_batch
    k = f"g{m['game']}_"
    noise, uniforms = _GZ[k + "noise"], _GZ[k + "uniforms"]
    size, A, G = m["size"], m["size"] ** 2, 3
    replay = azk.DeviceReplay(3 * m["buffer_len"] + 16, 2, size, size, A)
    res = self_play_batch("gomoku", lambda x: fixture_logits_value(x, A, m["variant"]), G, m["n_sims"], size=size,
                          noise_fn=lambda mv: np.tile(noise[min(mv, len(noise) - 1)], (G, 1)),
                          uniform_fn=lambda mv: np.full(G, uniforms[mv] if mv < len(uniforms) else 0.5), replay=replay)
    assert replay.size() == G * m["buffer_len"]
    states, pis, zs = replay.states.cpu().numpy(), replay.pis.cpu().numpy(), replay.zs.cpu().numpy()
    bases = sorted(r.replay_base for r in res)
    assert bases == [0, m["buffer_len"], 2 * m["buffer_len"]]
    for r in res:
        b = r.replay_base
        tuples = [(states[b + t], pis[b + t], float(zs[b + t])) for t in range(m["buffer_len"])]
        assert ro.digest(tuples) == m["buffer_digest"]                      # == the reference's ReplayBuffer contents
        want = ro.emit_tuples(r.boards, r.pis, r.winner)                    # and the oracle on the engine's own game record
        assert ro.digest(want) == m["buffer_digest"]


def test_ring_overwrites_oldest_and_sampling():
    import azk
    from selfplay import self_play_batch
    A = 49
    replay = azk.DeviceReplay(100, 2, 7, 7, A)                               # smaller than one batch's output
    res = self_play_batch("gomoku", lambda x: fixture_logits_value(x, A, "hash"), 8, 32, size=7, seed=1, replay=replay)
    total = sum((len(r.cells) if len(r.cells) <= 2 else 2 + 8 * (len(r.cells) - 2)) for r in res)
    assert int(replay.cursor.item()) == total and replay.size() == 100       # deque(maxlen) semantics (replay_buffer.py:10)
    s, p, z = replay.sample(64)
    assert s.shape == (64, 2, 7, 7) and p.shape == (64, A) and z.shape == (64, 1) and p.dtype == torch.float32
    assert torch.all((s == 0) | (s == 1)) and torch.allclose(p.sum(1), torch.ones(64, device=p.device), atol=1e-5)
    assert set(np.unique(z.cpu().numpy()).tolist()) <= {-1.0, 0.0, 1.0}


def test_stream_index_past_2_to_31():
    """The tuple stream index is 64-bit end to end: a ring whose cursor has passed 2^31 (hours of self-play) keeps
    receiving tuples at slot (index % capacity) and reports 64-bit bases (a 32-bit base went negative and the
    ring silently stopped filling)."""
    import azk
    from selfplay import self_play_batch
    A, cap = 49, 1000
    start = (1 << 31) - 5
    replay = azk.DeviceReplay(cap, 2, 7, 7, A)
    replay.cursor.fill_(start)
    replay.zs.fill_(7.0)                                                      # sentinel: every written slot gets z in {-1, 0, 1}
    res = self_play_batch("gomoku", lambda x: fixture_logits_value(x, A, "hash"), 4, 32, size=7, seed=2, replay=replay)
    total = sum((len(r.cells) if len(r.cells) <= 2 else 2 + 8 * (len(r.cells) - 2)) for r in res)
    assert total < cap
    assert int(replay.cursor.item()) == start + total
    bases = sorted(r.replay_base for r in res)
    assert bases[0] == start and bases[-1] > (1 << 31) and all(b >= start for b in bases)
    zs = replay.zs.cpu().numpy()
    written = {(start + t) % cap for t in range(total)}
    assert all((zs[i] != 7.0) == (i in written) for i in range(cap))
    from oracle import replay_oracle as ro
    states, pis = replay.states.cpu().numpy(), replay.pis.cpu().numpy()
    for r in res:
        n = len(r.cells) if len(r.cells) <= 2 else 2 + 8 * (len(r.cells) - 2)
        got = [(states[(r.replay_base + t) % cap], pis[(r.replay_base + t) % cap], float(zs[(r.replay_base + t) % cap])) for t in range(n)]
        assert ro.digest(got) == ro.digest(ro.emit_tuples(r.boards, r.pis, r.winner))


def test_sample_more_than_held_raises():
    import azk
    replay = azk.DeviceReplay(64, 2, 7, 7, 49)
    replay.add(np.zeros((2, 7, 7), np.float32), np.full(49, 1 / 49), [1.0])
    with pytest.raises(ValueError):                                           # np.random.choice(replace=False) raises too (replay_buffer.py:16)
        replay.sample(8)


def test_load_pickle_refuses_foreign_globals(tmp_path):
    import pickle
    import azk
    replay = azk.DeviceReplay(8, 2, 7, 7, 49)
    bad = tmp_path / "bad.pkl"
    with open(bad, "wb") as fh:
        pickle.dump({"x": print}, fh)                                         # a global outside the replay format
    with pytest.raises(pickle.UnpicklingError):
        replay.load_pickle(str(bad))


def test_rectangular_board_is_rejected():
    import azk
    eng = azk.Engine("connect4", 2, 8)
    replay = azk.DeviceReplay(16, 3, 6, 7, 7)
    with pytest.raises(azk.AzkError):
        eng.emit_finished(replay)


def test_collect_train_promote_loop():
    """main.start_train_loop's shape end to end on one GPU: games -> device replay ring -> reference loss / Adam ->
    new weights promoted, eval cache cleared, next collection runs with the new net."""
    from pvnet import NetConfig, init_weights
    from helper_train_loop import train_loop
    cfg = NetConfig(7, 7, 2, 49, patch_size=5, embed_dim=128, num_heads=4, depth=1)
    w0 = init_weights(cfg, 0)
    logs = []
    w1, hist = train_loop("gomoku", cfg, w0, iterations=2, games_per_iteration=16, n_sims=32, batch_size=64, size=7,
                          buffer_size=5000, cache_entries=512, log=logs.append)
    assert len(hist) == 2 and hist[1]["games"] >= 32 and hist[1]["buffer"] > hist[0]["buffer"] > 64
    assert all(np.isfinite(h["losses"]).all() for h in hist)
    changed = sum(float((w1[k] - w0[k]).abs().max()) > 0 for k in w0)
    assert changed >= len(w0) - 1
    # policy loss starts near log(49) for a random net and the L2 term is the dominant, slowly shrinking part
    assert 2.0 < hist[0]["losses"][1] < 6.0


def test_device_ring_interchanges_with_the_reference_buffer_format(tmp_path):
    """DeviceReplay <-> replay_buffer.ReplayBuffer: chronological export as the reference's deque of
    (state f32, pi f64, [z]) tuples (digest == the reference's own buffer), save / load in its on-disk format, host add."""
    import pickle
    import azk
    from oracle import replay_oracle as ro
    from selfplay import self_play_batch
    m = _META[0]
    k = f"g{m['game']}_"
    noise, uniforms = _GZ[k + "noise"], _GZ[k + "uniforms"]
    size, A = m["size"], m["size"] ** 2
    replay = azk.DeviceReplay(m["buffer_len"] + 8, 2, size, size, A)
    self_play_batch("gomoku", lambda x: fixture_logits_value(x, A, m["variant"]), 1, m["n_sims"], size=size,
                    noise_fn=lambda mv: noise[min(mv, len(noise) - 1)][None], uniform_fn=lambda mv: np.full(1, uniforms[mv] if mv < len(uniforms) else 0.5),
                    replay=replay)
    dq = replay.to_reference_deque()
    assert len(dq) == m["buffer_len"] and dq.maxlen == m["buffer_len"] + 8
    assert dq[0][0].dtype == np.float32 and dq[0][1].dtype == np.float64 and isinstance(dq[0][2], list)
    assert ro.digest([(s, p, z[0]) for s, p, z in dq]) == m["buffer_digest"]
    path = str(tmp_path / "replay_buffers" / "replay_buffer_test.pkl")
    replay.save_pickle(path)
    with open(path, "rb") as fh:
        raw = pickle.load(fh)                                             # a plain deque, as ReplayBuffer.load_pickle expects
    assert type(raw).__name__ == "deque" and len(raw) == m["buffer_len"]
    other = azk.DeviceReplay(m["buffer_len"] + 8, 2, size, size, A)
    other.load_pickle(path)
    assert other.size() == m["buffer_len"]
    assert ro.digest([(s, p, z[0]) for s, p, z in other.to_reference_deque()]) == m["buffer_digest"]
    # host-side add wraps like deque(maxlen)
    for j in range(10):
        other.add(np.full((2, size, size), j, np.float32), np.full(A, 1.0 / A), [1.0 if j % 2 else -1.0])
    dq2 = other.to_reference_deque()
    assert len(dq2) == m["buffer_len"] + 8 and float(dq2[-1][0][0, 0, 0]) == 9.0 and dq2[-1][2] == [1.0]
    assert ro.digest([(s, p, z[0]) for s, p, z in list(dq2)[:m["buffer_len"] - 2]]) == ro.digest([(s, p, z[0]) for s, p, z in list(dq)[2:]])
