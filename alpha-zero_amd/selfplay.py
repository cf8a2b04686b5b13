"""Batched self-play on the azk engine: thousands of <Game>.self_play loops as one lock-step batch.

Mirrors the per-move loop of games/gomoku.py:123-164 (and tictactoe.py:99-133, connect4.py:117-151):
fresh root -> n_sims simulations -> pi, q, raw board recorded -> sample (early moves) or most-visited
child -> make_move -> check_winner / draw.  Every game does that in the same step of the same kernels.

self_play_batch returns, per game, the tuple the reference's Gomoku.self_play returns
(boards, actions, policy_distributions, qs, winner) so train.collect_data's consumer
(train.save_data_to_buffer, train.py:30-49) can take it unchanged.
"""
import numpy as np

from azk import Engine

# gomoku.py:144 samples while move_count < 8; tictactoe.py:117 / connect4.py:135 sample every move
SAMPLE_UNTIL = {"gomoku": 8, "tictactoe": 1 << 30, "connect4": 1 << 30}


def cells_to_board(cells, planes, rows, cols, side_to_move):
    """int8 cell codes -> the reference's float32 board [F,R,C] (plane 2 = side to move, tictactoe.py:17,41)."""
    b = np.zeros((planes, rows, cols), np.float32)
    c = np.asarray(cells).reshape(rows, cols)
    b[0] = c == 1
    b[1] = c == 2
    if planes == 3:
        b[2] = side_to_move
    return b


class SelfPlayResult:
    __slots__ = ("boards", "actions", "pis", "qs", "winner", "cells")

    def __init__(self):
        self.boards, self.actions, self.pis, self.qs, self.winner, self.cells = [], [(-1, -1)], [], [], None, []

    def as_reference_tuple(self):
        """(boards, actions, policy_distributions, qs, winner) - gomoku.py:164"""
        return self.boards, self.actions, self.pis, self.qs, self.winner


def self_play_batch(game, evaluator, n_games, n_sims, size=None, seed=0, first_global_game=0, dirichlet=True,
                    alpha=0.03, noise_fn=None, uniform_fn=None, device=0, leaf_dtype="float32", engine=None,
                    max_moves=None, sample_until=None, stats=None):
    """Play n_games games to the end in one batch.

    evaluator(boards[n,F,R,C] CUDA) -> (logits [n,A], values [n] | [n,1]).
    RNG: by default Dirichlet noise and sampling uniforms come from the engine's counter-based generator
    keyed by (seed, first_global_game + g, move) - independent of how games are sharded over GPUs.
    noise_fn(move_idx) -> float64 [G, A] and uniform_fn(move_idx) -> float64 [G] (numpy) override it
    (that is how parity tests inject the reference's recorded np.random draws).
    """
    import torch
    eng = engine or Engine(game, n_games, n_sims, size=size, device=device, leaf_dtype=leaf_dtype)
    assert eng.G == n_games
    G, A = eng.G, eng.action_dim
    eng.reset_games()
    results = [SelfPlayResult() for _ in range(G)]
    active = np.ones(G, bool)
    su = SAMPLE_UNTIL[game] if sample_until is None else sample_until
    move = 0
    while active.any():
        if noise_fn is not None:
            nz = noise_fn(move)
            noise = torch.from_numpy(np.ascontiguousarray(nz, np.float64)).to(eng.device) if nz is not None else None
            uni = None
        else:
            noise, uni = eng.gen_noise(seed, first_global_game, move, alpha, want_noise=dirichlet)
        if uniform_fn is not None:
            uni = torch.from_numpy(np.ascontiguousarray(uniform_fn(move), np.float64)).to(eng.device)
        eng.search(evaluator, n_sims, noise if dirichlet else None)
        pi, q, _ = eng.root_stats()
        cells_before, to_move, _ = eng.get_positions()
        chosen, winner, done = eng.advance(uni, su)
        pi_h, q_h = pi.cpu().numpy(), q.cpu().numpy()
        chosen_h, winner_h, done_h = chosen.cpu().numpy(), winner.cpu().numpy(), done.cpu().numpy()
        for g in np.nonzero(active)[0]:
            r = results[g]
            r.boards.append(cells_to_board(cells_before[g], eng.planes, eng.rows, eng.cols, to_move[g]))
            r.pis.append(pi_h[g].copy())
            r.qs.append(float(q_h[g]))
            c = int(chosen_h[g])
            r.cells.append(c)
            r.actions.append((c // eng.cols, c % eng.cols))
            if done_h[g]:
                r.winner = int(winner_h[g])
                active[g] = False
        move += 1
        if max_moves is not None and move >= max_moves:
            break
    eng.check_error()
    if stats is not None:
        stats.update(eng.counters())
        stats["moves"] = move
    return results
