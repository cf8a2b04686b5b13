"""MCTS.mcts with the reference's signature (ai/mcts.py:11), executed by the HIP engine.

    MCTS.mcts(model, board, root, Game, mcts_iterations, dirichlet=True) -> None

Results come back the way the reference returns them: by mutation of `root` (visit, value, children
with visit / value / prior / prevAction); `board` is left exactly as passed.  `model` is any callable
tensor[n,F,R,C] (CUDA, float32) -> (logits[n,A], value[n,1]).  The Dirichlet draw is taken from the global
np.random stream at the same point the reference takes it (one np.random.dirichlet per search, utils.py:24),
so a seeded caller sees the same noise.

`MCTS.mcts_batch` is the batched form the self-play driver uses (many roots, one call).
"""
import numpy as np

from .node import Node

_ENGINES = {}
FACADE_CACHE_ENTRIES = 1 << 18      # eval-cache entries per game of the facade's engines (MCTS.cache; exact while nothing is evicted)


def _engine(Game, n_games, n_sims):
    import azk
    key = (Game.engine_name, Game.rows, Game.cols, n_games)
    eng = _ENGINES.get(key)
    if eng is None or eng.max_sims < n_sims:
        if eng is not None:
            eng.close()
        size = Game.rows if Game.engine_name == "gomoku" else None
        eng = azk.Engine(Game.engine_name, n_games, max(n_sims, 64), size=size,
                         cache_entries=FACADE_CACHE_ENTRIES if n_games <= 16 else 0)
        eng._facade_hits = 0
        _ENGINES[key] = eng
    return eng


class _EvalCacheView(dict):
    """MCTS.cache (mcts.py:7): the reference's process-global dict position -> (policy, value), which every search of every
    model reads and fills until main.py:55 clears it.  Here it is the eval cache of the facade's engines (keyed by the exact
    position, shared by whatever models are searched, persistent across calls); clear() is the operation callers use."""

    def clear(self):
        for eng in _ENGINES.values():
            if eng.cache_entries:
                eng.clear_cache()
        super().clear()


def _cells(board):
    return (np.asarray(board[0]) == 1).astype(np.int8) + 2 * (np.asarray(board[1]) == 1).astype(np.int8)


class MCTS:
    # ai/mcts.py:7-9: cache hits are counted in `matched`, simulations in `mcts_count`, as the reference does
    cache = _EvalCacheView()
    matched = 0
    mcts_count = 0

    @staticmethod
    def mcts(model, board, root, Game, mcts_iterations, dirichlet=True):
        if model is None:
            MCTS.mcts_vanilla(board, root, Game, mcts_iterations)
            return
        MCTS.mcts_batch(model, [board], [root], Game, mcts_iterations, dirichlet)

    @staticmethod
    def mcts_vanilla(board, root, Game, mcts_iterations):
        """model=None (mcts.py:57-79): UCB1 walk, expansion without priors, uniform-random rollouts - all inside the
        engine.  The rollouts draw np.random.randint from the GLOBAL np.random stream in the reference; here the
        global MT19937 state is handed to the device generator and written back afterwards, so a seeded caller gets
        the reference's draws and finds np.random exactly where the reference would have left it."""
        import azk
        eng = _engine(Game, 1, mcts_iterations)
        eng.set_positions(_cells(board).reshape(1, -1), [root.currentPlayer], [root.move_count])
        st = np.random.get_state()
        eng.vanilla_set_rng(azk.mt_state_from_numpy(st)[None])
        eng.vanilla_search(mcts_iterations, chunk=64 if Game.rows * Game.cols > 64 else None)
        eng.check_error()
        np.random.set_state(azk.mt_state_to_numpy(eng.vanilla_get_rng()[0], st))
        MCTS.mcts_count += mcts_iterations
        MCTS._materialise(eng, [root], Game, prior_none=True)

    @staticmethod
    def mcts_batch(model, boards, roots, Game, mcts_iterations, dirichlet=True, noise=None):
        import torch
        G = len(boards)
        eng = _engine(Game, G, mcts_iterations)
        A = Game.action_dim
        cells = np.stack([_cells(b).reshape(-1) for b in boards])
        eng.set_positions(cells, [r.currentPlayer for r in roots], [r.move_count for r in roots])
        nz = None
        if dirichlet:
            if noise is None:
                noise = np.stack([np.random.dirichlet([0.03] * A) for _ in range(G)])     # utils.py:12,24
            nz = torch.from_numpy(np.ascontiguousarray(noise, np.float64)).to(eng.device)

        def evaluator(x):
            logits, value = model(x)
            return logits, value
        eng.search(evaluator, mcts_iterations, nz)
        eng.check_error()
        if eng.cache_entries:
            hits = eng.counters()["cache_hits"]
            MCTS.matched += hits - eng._facade_hits
            eng._facade_hits = hits
        MCTS.mcts_count += mcts_iterations * G
        MCTS._materialise(eng, roots, Game, f32_prior=not dirichlet)

    @staticmethod
    def _materialise(eng, roots, Game, f32_prior=False, prior_none=False):
        _, q, rv = eng.root_stats()
        q, rv = q.cpu().numpy(), rv.cpu().numpy()
        for g, root in enumerate(roots):
            ch = eng.root_children(g)
            root.visit = int(rv[g])
            root.value = float(eng.export_tree(g, cap=1)["value"][0])      # W of the root, exact
            root.children = []
            for cell, n, w, p in zip(ch["cell"], ch["visit"], ch["value"], ch["prior"]):
                prior = 0.0 if prior_none else (np.float32(p) if f32_prior else np.float64(p))   # node.py:21 default prior
                node = Node(root, (int(cell) // Game.cols, int(cell) % Game.cols), 1 - root.currentPlayer,
                            root.move_count + 1, prior)
                node.visit = int(n)
                node.value = float(w)
                root.children.append(node)
