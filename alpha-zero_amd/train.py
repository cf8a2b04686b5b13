"""train.py of the reference (train.py:8-123) with its names and signatures, on the batched engine.

    collect_data(Game, model, buffer, iterations, mcts_iter, display=False)   train.py:54-83
    save_data_to_buffer(Game, buffer, data)                                   train.py:30-49
    rotate_data / flip_data                                                   train.py:8-27
    train(model, batch_size, buffer, train_iterations, lr, device)            train.py:85-123

`buffer` is anything with the reference's ReplayBuffer interface (add / sample / size) - the reference's own class or
azk.DeviceReplay.  With a DeviceReplay, collect_data plays all `iterations` games as ONE lock-step batch and the engine
emits the (state, pi, z) tuples on the device (azk_emit_finished); with a host buffer the games still run as one batch and
the tuples are appended through save_data_to_buffer exactly as the reference does it, game by game.
`model`: pvnet.PolicyValueNet (or, for collect_data, any callable tensor[n,F,R,C] -> (logits[n,A], value[n,1])).
"""
import numpy as np


def rotate_data(board, policy, k=0):
    """train.py:8-15: np.rot90 on the board planes and on the policy reshaped to the board."""
    rows, cols = board.shape[-2], board.shape[-1]
    return np.rot90(board, k=k, axes=(1, 2)), np.rot90(policy.reshape(rows, cols), k=k).reshape(-1)


def flip_data(board, policy, mode="lr"):
    """train.py:17-27: left-right / top-bottom flips of the planes and of the policy."""
    rows, cols = board.shape[-2], board.shape[-1]
    p = policy.reshape(rows, cols)
    if mode == "lr":
        return np.flip(board, axis=2), np.fliplr(p).reshape(-1)
    return np.flip(board, axis=1), np.flipud(p).reshape(-1)


def save_data_to_buffer(Game, buffer, data):
    """train.py:30-49: z = +reward for the winner's positions, -reward for the loser's; canonical boards; positions 0 and 1
    once, every later position with its 8 dihedral images in the reference's order."""
    boards, actions, policies, qs, winner, reward = data
    current = 0
    for i in range(len(boards)):
        target = [reward] if current == winner else [-reward]
        canon = Game.get_canonical_board(boards[i], current)
        current = 1 - current
        if i < 2:
            buffer.add(canon, policies[i], target)
            continue
        for r in range(4):
            b, p = rotate_data(canon, policies[i], k=r)
            buffer.add(b, p, target)
            if r < 2:
                for mode in ("lr", "tb"):
                    fb, fp = flip_data(b, p, mode)
                    buffer.add(fb, fp, target)


def collect_data(Game, model, buffer, iterations, mcts_iter, display=False, seed=None, batched=True):
    """train.py:54-83: `iterations` self-play games of Game with `model` into `buffer`; returns [first, second, draw] counts.
    batched=False plays the games one after the other through Game().self_play and save_data_to_buffer exactly as the
    reference does (global np.random stream: a seeded caller gets the reference's buffer); the default plays them as one
    engine batch with the engine's own counter-based RNG."""
    from azk import DeviceReplay
    from selfplay import self_play_batch
    if not batched:
        results = [0, 0, 0]
        for _ in range(iterations):
            game = Game()
            boards, actions, pis, qs, winner = game.self_play(model, mcts_iter, display)
            results[2 if winner == -1 else winner] += 1
            save_data_to_buffer(Game, buffer, (boards, actions, pis, qs, winner, 0 if winner == -1 else 1))
        return results
    if seed is None:
        seed = int(np.random.randint(0, 2 ** 31 - 1))
    on_device = isinstance(buffer, DeviceReplay)
    leaf_dtype = "bfloat16" if getattr(model, "dtype", None) is not None and str(model.dtype).endswith("bfloat16") else "float32"
    res = self_play_batch(Game.engine_name, model, iterations, mcts_iter, size=Game._size(), seed=seed, leaf_dtype=leaf_dtype,
                          replay=buffer if on_device else None)
    results = [0, 0, 0]
    for r in res:
        results[2 if r.winner == -1 else r.winner] += 1
        if not on_device:
            reward = 0 if r.winner == -1 else 1
            save_data_to_buffer(Game, buffer, (r.boards, r.actions, r.pis, r.qs, r.winner, reward))
        if display:
            Game.display_board(r.boards[-1])
    return results


def train(model, batch_size, buffer, train_iterations, lr, device):
    """train.py:85-123 for a pvnet.PolicyValueNet: a fresh Adam, `train_iterations` sampled batches, the reference's loss;
    the network's weights are updated in place; returns (loss, policy_loss, value_loss, l2) of the last iteration."""
    import torch
    from trainer import Trainer
    tr = Trainer(model.cfg, model.state_dict(), device=device)

    def batches():
        for _ in range(train_iterations):
            s, p, z = buffer.sample(batch_size)
            yield s, p.to(torch.float32), z.reshape(-1, 1)
    dist = torch.distributed if torch.distributed.is_available() and torch.distributed.is_initialized() else None
    out = tr.train(batches(), lr, dist=dist)
    model.load_state_dict(tr.state_dict())
    return out
