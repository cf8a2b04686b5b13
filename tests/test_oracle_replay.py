"""(state, pi, z) emission: the numpy oracle against the digests recorded from the reference's
train.save_data_to_buffer on whole games (tests/golden/games.npz).  CPU only."""
import numpy as np

from conftest import golden_meta, load_golden
from oracle import replay_oracle as ro

_Z = load_golden("games.npz")
_META = [m for m in golden_meta(_Z) if "buffer_digest" in m]


def boards_of(m, k):
    size = m["size"]
    cells = _Z[k + "board_cells"]
    out = []
    for c in cells:
        b = np.zeros((2, size, size), np.float32)
        c = c.reshape(size, size)
        b[0], b[1] = c == 1, c == 2
        out.append(b)
    return out


def test_emission_matches_reference_buffers():
    assert len(_META) == 6                                           # four network-mode Gomoku games + two vanilla ones
    for m in _META:
        k = f"g{m['game']}_"
        tuples = ro.emit_tuples(boards_of(m, k), list(_Z[k + "pis"]), m["winner"])
        assert len(tuples) == m["buffer_len"]
        assert ro.digest(tuples) == m["buffer_digest"]
