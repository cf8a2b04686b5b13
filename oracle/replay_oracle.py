"""CPU restatement (numpy) of the reference's (state, pi, z) emission - TEST INFRASTRUCTURE.

train.collect_data -> train.save_data_to_buffer (train.py:30-49, rotate_data :8-15, flip_data :17-27):
for position i of a finished game (side to move = i mod 2):
    z      = +reward if side == winner else -reward          (reward = 0 for a draw, 1 otherwise; train.py:71-75)
    state  = Game.get_canonical_board(boards[i], side)
    positions 0 and 1 are stored once; every other position 8 times, in this order:
        rot0, lr(rot0), tb(rot0), rot90, lr(rot90), tb(rot90), rot180, rot270      (np.rot90 is counter-clockwise)
Pinned by the sha256 digests recorded from the reference in tests/golden/games.npz (buffer_digest).
"""
import hashlib
import struct

import numpy as np


def canonical(board, side):
    if side == 0:
        return board
    out = np.empty_like(board)
    out[0], out[1] = board[1], board[0]
    if board.shape[0] == 3:
        out[2] = board[2]
    return out


def emit_tuples(boards, pis, winner):
    """-> list of (state float32 [F,R,C], pi float64 [A], z float) in the reference's buffer order."""
    reward = 0 if winner == -1 else 1
    out = []
    for i, (b, p) in enumerate(zip(boards, pis)):
        side = i & 1
        z = float(reward if side == winner else -reward)
        st = canonical(b, side)
        rows, cols = st.shape[1], st.shape[2]
        if i < 2:
            out.append((st, p, z))
            continue
        for r in range(4):
            br = np.rot90(st, k=r, axes=(1, 2)).copy()
            pr = np.rot90(p.reshape(rows, cols).copy(), k=r)
            out.append((br, pr.flatten(), z))
            if r < 2:
                p2 = pr.copy()
                out.append((np.flip(br, axis=2).copy(), np.flip(p2, axis=1).copy().flatten(), z))
                out.append((np.flip(br, axis=1).copy(), np.flip(p2, axis=0).copy().flatten(), z))
    return out


def digest(tuples):
    h = hashlib.sha256()
    for state, pi, z in tuples:
        h.update(np.ascontiguousarray(state, np.float32).tobytes())
        h.update(np.ascontiguousarray(pi, np.float64).tobytes())
        h.update(struct.pack("<d", float(z)))
    return h.hexdigest()
